"""GPU parity of every C-ABI entry point against a plain fp64 torch restatement of the same op
(tests/gpu_probe.py holds the cases: forward and backward, fp32 at ~1e-5 and bf16 at ~1e-2 relative to
the tensor scale, ragged M/N/K tails, PatchMerging gather, shifted windows with masks, packed and
separate k/v, dropout statistics, BatchNorm train/eval)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(fn_name):
    from tests import gpu_probe as P
    P.RES.clear()
    P.section(getattr(P, fn_name))
    bad = [n for n, ok in P.RES if not ok]
    assert P.RES and not bad, f"{fn_name}: failed cases: {bad}"


@pytest.mark.parametrize("case", ["t_linear", "t_linear_large", "t_wgrad", "t_wgrad_large", "t_mlp_fused", "t_layernorm", "t_wattn", "t_mha", "t_misc"])
def test_op_parity(case):
    assert torch.cuda.is_available()
    _run(case)


def test_argument_errors_raise_on_gpu():
    from facialmmt_amd import _lib, ops
    dev = torch.device("cuda:0")
    with pytest.raises(_lib.FmmtError, match="EINVAL"):
        ops.linear_raw(torch.zeros(8, 100, device=dev, dtype=torch.bfloat16), torch.zeros(96, 100, device=dev, dtype=torch.bfloat16), None)
    with pytest.raises(_lib.FmmtError):
        ops.mha_core(torch.zeros(4, 1, 500, device=dev), torch.zeros(4, 1, 1000, device=dev), None, 4, 1.0)
    with pytest.raises(_lib.FmmtError, match="unsupported activation dtype"):
        ops.layer_norm(torch.zeros(4, 96, device=dev, dtype=torch.float16), torch.ones(96, device=dev), torch.zeros(96, device=dev))


def test_register_staged_weight_gradient_kernel_on_the_large_shapes():
    """the many-token stage-2/3 weight gradients take linear_tn_dma_kernel (256x256 / 192x384 tiles) by default -- covered by
    gpu_probe.t_wgrad_large in the main run; FMMT_TN_DMA=0 sends the same shapes through the register-staged linear_tn_kernel
    (the path every launch with a DropPath scale or a recomputed activation still takes).  Own process: the switch is read once."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from tests import gpu_probe as P\n"
            "P.section(P.t_wgrad_large)\n"
            "bad = [n for n, ok in P.RES if not ok]\n"
            "print('CASES', len(P.RES), 'FAILED', bad)\n"
            "sys.exit(1 if bad or not P.RES else 0)\n") % root
    env = dict(os.environ, FMMT_TN_DMA="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_non_default_gemm_switches_stay_correct():
    """The A/B switches that are off by default keep their code paths: the persistent NT kernel with the epilogue operand
    prefetched into registers for GELU' launches too (FMMT_NT_P256_OPS=2; 0: none), its GELU + pre-activation epilogue through the LDS scratch
    (FMMT_NT_P256_LDSGELU=1), compiler-scheduled fragment reads and direct epilogue stores (FMMT_NT_P256_BATCH=0,
    FMMT_NT_P256_LDSEPI=0: the older forms), 64-token steps and the older
    one-workgroup form for few-token weight gradients (FMMT_TN_FEW64=1, FMMT_TN_FEW=0), unscaled-only DMA weight gradients
    (FMMT_TN_DMA_SCALED=0), operand-free launches through the operand-prefetch instantiation (FMMT_NT_P256_PLAINOP=2), 64 x 64 / 64 x 128 tiles for the few-token NT launches (FMMT_NT_SMALL=2 / 0), weight rows in channel order in the persistent kernel's LDS stages (FMMT_NT_P256_WROWS=0).  One process per setting: the switches are read once."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from tests import gpu_probe as P\n"
            "P.section(P.t_linear)\n"
            "P.section(P.t_linear_large)\n"
            "P.section(P.t_wgrad)\n"
            "P.section(P.t_wgrad_large)\n"
            "bad = [n for n, ok in P.RES if not ok]\n"
            "print('CASES', len(P.RES), 'FAILED', bad)\n"
            "sys.exit(1 if bad or not P.RES else 0)\n") % root
    for extra in ({"FMMT_NT_P256_OPS": "2", "FMMT_NT_P256_LDSGELU": "1", "FMMT_TN_FEW64": "1", "FMMT_TN_FEW": "0", "FMMT_TN_DMA_SCALED": "0"},
                  {"FMMT_NT_P256_BATCH": "0", "FMMT_NT_P256_LDSEPI": "0", "FMMT_NT_P256_OPS": "0", "FMMT_NT_SLAB": "0", "FMMT_NT_WSLAB": "0", "FMMT_NT_SMALL": "0"},
                  {"FMMT_NT_P256_OPS": "0", "FMMT_NT_P256_PLAINOP": "2", "FMMT_NT_SMALL": "2", "FMMT_NT_P256_WROWS": "0"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, str(extra) + r.stdout[-2000:] + r.stderr[-2000:]
