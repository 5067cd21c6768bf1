"""profiles/traffic.json from the PMC summaries of a profile round: python tools/make_traffic_json.py DIR TAG
(DIR holds pmc_fetch.md / pmc_write.md as tools/summarize_prof.py wrote them).  Keys are bench.py's family names."""
import json, os, re, sys

d, tag = sys.argv[1], sys.argv[2]
# bench.py family name <- substring(s) of the rocprofv3 kernel name
FAMILIES = {
    "linear_tn_kernel<bf16,32>": ["16linear_tn_kernelIDF16bLi32ELb0"],
    "linear_nt_p256_kernel<192,64,2,true,false,true>": ["linear_nt_p256_kernel<192, 64, 2, true, false, true>"],
    "linear_nt_p256_kernel<192,64,2,true,true,false>": ["linear_nt_p256_kernel<192, 64, 2, true, true, false>"],
    "linear_nt_p256_kernel<256,64,2,true,false,true>": ["linear_nt_p256_kernel<256, 64, 2, true, false, true>"],
    "linear_nt_ph3_kernel<384x128>": ["linear_nt_ph3_kernel<2, true, 4>", "linear_nt_ph3_kernel<5, true, 4>", "linear_nt_ph3_kernel<6, true, 4>"],
    "linear_nt_ph3_kernel<192x256>": ["linear_nt_ph3_kernel<2, true, 2>", "linear_nt_ph3_kernel<5, true, 2>", "linear_nt_ph3_kernel<3, true, 2>", "linear_nt_ph3_kernel<6, true, 2>"],
    "linear_nt_ph_kernel<256x256>": ["linear_nt_ph_kernel<"],
    "linear_nt_deep32_kernel<0,128>": ["linear_nt_deep32_kernel<0, 128>"],
    "linear_nt_deep32_kernel<6,128>": ["linear_nt_deep32_kernel<6, 128>"],
    "linear_tn_dma_kernel<192,384,4,2,true>": ["linear_tn_dma_kernel<192, 384, 4, 2, true>"],
    "linear_tn_dma_kernel<192,384,4,2,false>": ["linear_tn_dma_kernel<192, 384, 4, 2, false>"],
    "linear_tn_dma_kernel<384,192,4,4,false>": ["linear_tn_dma_kernel<384, 192, 4, 4, false>"],
    "linear_tn_dma_kernel<256,256,4,2,false>": ["linear_tn_dma_kernel<256, 256, 4, 2, false>"],
    "linear_tn_dma_kernel<256,256,4,2,true>": ["linear_tn_dma_kernel<256, 256, 4, 2, true>"],
    "mlp_fused_fwd_kernel<LN><C=96>": ["mlp_fused_fwd_kernel<96, true"],
    "mlp_fused_fwd_kernel<LN><C=192>": ["mlp_fused_fwd_kernel<192, true"],
    "mlp_fused_bwd_kernel<LN'><C=96>": ["mlp_fused_bwd_kernel<96, true"],
    "mlp_fused_bwd_kernel<C=192>": ["mlp_fused_bwd_kernel<192"],
    "wattn_mfma_bwd_kernel<recompute><C=96>": ["wattn_mfma_bwd_kernel<0, 4, 96>", "wattn_mfma_bwd_kernel<1, 4, 96>"],
    "wattn_mfma_bwd_kernel": ["wattn_mfma_bwd_kernel<0, 2, 0>", "wattn_mfma_bwd_kernel<1, 2, 0>"],
    "wblock_fwd_kernel<C=96>": ["wblock_fwd_kernel<96, 8>"],
    "lin_lnbwd_kernel": ["lin_lnbwd_kernel<96, 288"],
    "adamw_batch_kernel": ["adamw_batch_kernel"],
    "grad_handover_kernel": ["grad_handover_kernel"],
}


def rows(path, counter):
    out = []
    for line in open(path):
        m = re.match(r"\| `(.*)` \| (\w+) \| (\d+) \| ([\d.e+]+) \| ([\d.e+]+) \|", line)
        if m and m.group(2) == counter:
            out.append((m.group(1), int(m.group(3)), float(m.group(4))))
    return out


fetch, write = rows(os.path.join(d, "pmc_fetch.md"), "FETCH_SIZE"), rows(os.path.join(d, "pmc_write.md"), "WRITE_SIZE")
res = {"_comment": "HBM-side traffic per launch of the largest kernel families from rocprofv3 PMC passes (own passes, --graphs 0): bytes = (2 * FETCH_SIZE + "
                   "WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> doubled, as MI355X_MICROARCH.md prescribes; "
                   "WRITE_SIZE uncalibrated; Infinity-Cache hits are counted, so this is L2-to-fabric traffic, an upper bound of HBM traffic). bench.py copies the "
                   "entry of the kernel it reports into roofline.traffic.  Written by tools/make_traffic_json.py.",
       "_source": f"round {tag.lstrip('r0') or tag} (tools/profile_round.sh {tag}): profiles/{tag}_pmc_fetch.md, profiles/{tag}_pmc_write.md"}
for fam, subs in FAMILIES.items():
    f = [(n, s) for name, n, s in fetch if any(x in name for x in subs)]
    w = [(n, s) for name, n, s in write if any(x in name for x in subs)]
    if not f or not w:
        continue
    nf, nw = sum(n for n, _ in f), sum(n for n, _ in w)
    res[fam] = {"fetch_size_kb": round(sum(s for _, s in f) / nf), "write_size_kb": round(sum(s for _, s in w) / nw), "dispatches": nf, "round": tag}
json.dump(res, sys.stdout, indent=2)
print()
