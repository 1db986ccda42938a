"""What a fused run's staging makes of fp32 rows (include/papr_hip.h: papr_f16_rows), restated in torch for the tests: the row times the power of two
of its maximum, rounded once to f16 (and, in the parity runs' split form, the rounded-off rest as a second f16)."""
import torch


def expected(rows, split):
    """rows (M, W) fp32 on the device -> (hi, lo or None, inv, scale, max) as the producer (papr_attn_tail_bwd) must write them."""
    mx = rows.abs().amax(1)
    e = torch.frexp(mx)[1] - 1                                          # max in [2^e, 2^(e + 1))
    if split:       # parity: the maximum into [2^13, 2^14); a zero row: scale 1
        e = torch.where(mx > 0, e, torch.full_like(e, 13))
        target = 13
    else:           # one-product data-gradient rows: into [2^3, 2^4) (h3_common.h: ONE_TARGET_E), the exponent clamped at -40 (ONE_EMIN_DGRAD)
        e = torch.where(mx > 0, torch.clamp(e, min=-40), torch.full_like(e, -40))
        target = 3
    scale = ((127 + target - e).to(torch.int32) << 23).view(torch.float32)         # 2^(target - e), exactly (torch.ldexp on the device is not)
    inv = ((127 - target + e).to(torch.int32) << 23).view(torch.float32)
    scaled = rows * scale[:, None]
    hi = scaled.to(torch.float16)
    lo = (scaled - hi.float()).to(torch.float16) if split else None
    return hi, lo, inv, scale, mx


def fill(dst, rows):
    """Fill an ops.F16Rows from fp32 rows (the form follows dst: split if it has a lo plane)."""
    hi, lo, inv, scale, mx = expected(rows, dst.lo is not None)
    dst.hi[:, :hi.shape[1]].copy_(hi)
    if lo is not None:
        dst.lo[:, :lo.shape[1]].copy_(lo)
    dst.tables[0].copy_(inv); dst.tables[1].copy_(scale); dst.tables[2].copy_(mx)
    return dst
