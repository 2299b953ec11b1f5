"""catan_linear_wgrad_big (csrc/catan_wgrad_big.hip): the weight gradient of the observation trunk's 992 -> 512 product
(RL/models/observation_module.py:58-63 in the backward of RL/ppo/ppo.py:66) against the fp32 product of the same bf16 operands, its bias
gradient, run-to-run bit-equality, and the autograd node that uses it against F.linear's own backward."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,I,O", [(70001, 992, 512), (204800, 992, 512), (16385, 264, 128), (40000, 1016, 256)])
def test_wgrad_big_equals_the_fp32_product(hip_lib, rows, I, O):
    from settlers_of_catan_rl_amd import nn_kernels
    g = torch.Generator(device="cuda").manual_seed(rows)
    x = torch.randn(rows, I, device="cuda", generator=g).to(torch.bfloat16)
    dy = (torch.randn(rows, O, device="cuda", generator=g) * 0.01).to(torch.bfloat16)
    assert nn_kernels.wgrad_big_supported(rows, I, O)
    dw, db = nn_kernels.wgrad_big(x, dy)
    ref = torch.zeros(O, I, dtype=torch.float64, device="cuda")
    for r0 in range(0, rows, 32768):                                      # (fp64 in chunks: the reference must not round)
        ref += dy[r0:r0 + 32768].double().t() @ x[r0:r0 + 32768].double()
    refb = dy.double().sum(0)
    assert dw.shape == (O, I) and db.shape == (O,)
    assert float((dw.double() - ref).norm() / ref.norm()) < 2e-6, float((dw.double() - ref).norm() / ref.norm())
    assert float((db.double() - refb).norm() / refb.norm()) < 2e-6
    dw2, db2 = nn_kernels.wgrad_big(x, dy)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)                   # the row groups' partials are added in index order


def test_linear_big_autograd_node_against_the_library_backward(hip_lib):
    from settlers_of_catan_rl_amd import nn_kernels
    g = torch.Generator(device="cuda").manual_seed(3)
    rows, I, O = 32768, 992, 512
    x = torch.randn(rows, I, device="cuda", generator=g).to(torch.bfloat16).requires_grad_(True)
    w = (torch.randn(O, I, device="cuda", generator=g) * 0.03).requires_grad_(True)
    b = torch.randn(O, device="cuda", generator=g).requires_grad_(True)
    gy = (torch.randn(rows, O, device="cuda", generator=g) * 0.01).to(torch.bfloat16)
    y1 = nn_kernels.linear_big(x, w, b)
    g1 = torch.autograd.grad(y1, (x, w, b), gy)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y0 = torch.nn.functional.linear(x, w, b)
    g0 = torch.autograd.grad(y0, (x, w, b), gy)
    assert torch.equal(y1, y0)
    assert torch.equal(g1[0], g0[0])                                       # dX: the same library product
    assert g1[1].dtype == torch.float32
    # the library's weight gradient leaves in bf16 (8 mantissa bits); the kernel's stays in fp32: compare both with the fp64 product
    ref = gy.double().t() @ x.detach().double()
    e1, e0 = float((g1[1].double() - ref).norm() / ref.norm()), float((g0[1].double() - ref).norm() / ref.norm())
    assert e1 < 2e-6 and e1 <= e0, (e1, e0)
    assert float((g1[2].double() - gy.double().sum(0)).abs().max()) < 1e-4
