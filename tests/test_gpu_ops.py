"""GPU parity of every C-ABI entry point against a plain fp64 torch restatement of the same op
(tests/support_op_cases.py holds the cases: forward and backward, fp32 at ~1e-5 and bf16 at ~1e-2 relative to
the tensor scale, ragged M/N/K tails, PatchMerging gather, shifted windows with masks, packed and
separate k/v, dropout statistics, BatchNorm train/eval)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(fn_name):
    from tests import support_op_cases as P
    P.RES.clear()
    P.section(getattr(P, fn_name))
    bad = [n for n, ok in P.RES if not ok]
    assert P.RES and not bad, f"{fn_name}: failed cases: {bad}"


@pytest.mark.parametrize("case", ["t_linear", "t_linear_large", "t_wgrad", "t_wgrad_large", "t_mlp_fused", "t_gelu_tail", "t_layernorm", "t_wattn", "t_mha", "t_misc"])
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
