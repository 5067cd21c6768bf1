// Window geometry and fragment helpers shared by the matrix-core window-attention kernels (wattn_mfma.hip: attention core on a
// materialised qkv; wblock.hip: the whole attention half of a Swin block per window).  roll / window_partition /
// window_reverse (Swin_Transformer.py:33-62,244,261) exist only as the address arithmetic below.
#pragma once
#include "fmmt_common.h"

namespace {

constexpr int TOK = 49, WS = 7, HD = 32;
constexpr int TP = 40;       // LDS tile pitch in bf16 (80 B rows: 16-byte aligned, spreads banks)
constexpr int BPM = 68;      // LDS bias pitch (floats): 272-byte rows -- at 64 the 16 query rows of an f32x4 bias read sat on the same 4 banks (round 5: SQ_LDS_BANK_CONFLICT 55 % of SQ_LDS_IDX_ACTIVE)
constexpr float NEG_BIG = -1.0e30f;

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ bf16x8 ld_frag(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }

__device__ __forceinline__ bf16x8 zero_frag() {
    bf16x8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = (bf16)0.f;
    return z;
}

// A-operand fragment of X^T for X stored [row][HD] in LDS: MFMA row i <-> channel (i>>2)*8 + dt*4 + (i&3),
// k-slot (g, e) <-> row r0 + e (e < 4), r0 + 16 + (e - 4) (e >= 4)   with r0 = 32*ks + 4*g
__device__ __forceinline__ bf16x8 tr_fragT(const bf16* tile, int r0, int dt, int li) {
    const bf16* a0 = tile + (r0 + (li >> 2)) * TP + (li & 3) * 8 + dt * 4;
    union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
    u.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
    u.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 16 * TP));
    return u.v;
}

__device__ __forceinline__ bf16x8 pack8(const float* lo4, const float* hi4) {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = (bf16)lo4[e]; v[4 + e] = (bf16)hi4[e]; }
    return v;
}

__device__ __forceinline__ float xor_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float xor_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

struct LaneGeom {
    int di[4], dj[4];        // window-local (row, col) of slot t*16 + li (clamped to slot 48)
    bool valid[4];
    unsigned nearH, nearW;   // bit (t*4 + r): slot 16t + 4g + r lies in the "near" part (i < 7 - shift) -- std mask
    unsigned ownH, ownW;     // bit t: own slot 16t + li in the near part
};

__device__ __forceinline__ LaneGeom lane_geom(int li, int lg, int shift) {
    LaneGeom G;
    G.nearH = G.nearW = G.ownH = G.ownW = 0u;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int slot = t * 16 + li;
        G.valid[t] = slot < TOK;
        const int cs = slot < TOK ? slot : TOK - 1;
        G.di[t] = cs / WS;
        G.dj[t] = cs - G.di[t] * WS;
        if (G.di[t] < WS - shift) G.ownH |= 1u << t;
        if (G.dj[t] < WS - shift) G.ownW |= 1u << t;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int s2 = t * 16 + lg * 4 + r;
            s2 = s2 < TOK ? s2 : TOK - 1;
            const int i2 = s2 / WS, j2 = s2 - i2 * WS;
            if (i2 < WS - shift) G.nearH |= 1u << (t * 4 + r);
            if (j2 < WS - shift) G.nearW |= 1u << (t * 4 + r);
        }
    }
    return G;
}

struct WinPos { int img, wy, wx, w; bool lastrow, lastcol; };

template <class A>
__device__ __forceinline__ WinPos win_pos(const A& p, int b_) {
    const int nWx = p.W / WS, nWy = p.H / WS, nW = nWx * nWy;
    WinPos P;
    P.img = b_ / nW;
    P.w = b_ - P.img * nW;
    P.wy = P.w / nWx;
    P.wx = P.w - P.wy * nWx;
    P.lastrow = P.wy == nWy - 1;
    P.lastcol = P.wx == nWx - 1;
    return P;
}

template <class A>
__device__ __forceinline__ size_t tok_of(const A& p, const WinPos& P, int di, int dj) {
    int hh = P.wy * WS + di + p.shift;
    if (hh >= p.H) hh -= p.H;
    int ww = P.wx * WS + dj + p.shift;
    if (ww >= p.W) ww -= p.W;
    return (size_t)P.img * p.H * p.W + (size_t)hh * p.W + ww;
}

// 16-bit mask of "other" slots (bit t*4+r <-> slot 16t+4g+r) that sit in a different shift-region than
// the lane's own slot of tile `t_own`
__device__ __forceinline__ unsigned std_mask_bits(const LaneGeom& G, const WinPos& P, int t_own) {
    unsigned m = 0u;
    if (P.lastrow) m |= ((G.ownH >> t_own) & 1u) ? ~G.nearH : G.nearH;
    if (P.lastcol) m |= ((G.ownW >> t_own) & 1u) ? ~G.nearW : G.nearW;
    return m & 0xFFFFu;
}

}  // namespace
