"""Which form of the MoE block a token count gets (ops/moe.py), checked on the two layer shapes the limits were measured
on.  Pure host logic: runs without a GPU."""
from triton_distributed_gh_amd.ops import moe


def _form(n, topk, E, H, Ip):
    """The w == 1 dispatch of layers/tp_moe.py: TP_MoE.dist_triton_fwd, shape checks aside."""
    big = 3 * H * Ip * 2 > moe.MOE_GROUP16_BIG_EXPERT_BYTES
    mid_limit = (moe.MOE_GROUP_MID_ROWS_PER_BIG_EXPERT if big else moe.MOE_GROUP_MID_ROWS_PER_EXPERT) * E
    mid = n * topk <= mid_limit and moe.moe_group_mid_preferred(n, topk, E, H, Ip)
    decode_ok = n <= moe.MOE_DECODE_MAX_TOKENS or (n <= moe.MOE_GROUP16_MAX_TOKENS
                                                  and n * topk <= moe.group16_row_limit(E, H, Ip))
    if not mid and decode_ok:
        return "group16" if moe._group16_takes(n, topk, E, Ip, H) else "pairs"
    return "mid" if mid else "gemm256"


def test_qwen3_30b_a3b_layer_forms():
    shape = (8, 128, 2048, 768)           # top-8 of 128 experts of 9.4 MB
    assert [_form(n, *shape) for n in (1, 4, 8)] == ["pairs"] * 3
    assert [_form(n, *shape) for n in (9, 16, 17, 64, 128, 240)] == ["group16"] * 6
    assert [_form(n, *shape) for n in (241, 256, 512, 1024, 1600)] == ["mid"] * 5
    assert [_form(n, *shape) for n in (1601, 2048, 8192)] == ["gemm256"] * 3


def test_mixtral_layer_forms():
    shape = (2, 8, 4096, 14336)           # top-2 of 8 experts of 352 MB: grouped from the first token on
    assert [_form(n, *shape) for n in (1, 2, 16, 48)] == ["group16"] * 4
    assert [_form(n, *shape) for n in (49, 64, 128, 320)] == ["mid"] * 4
    assert [_form(n, *shape) for n in (321, 512, 4096)] == ["gemm256"] * 3


def test_block_height_rules():
    # sixteen-row form: blocks of 32 rows only past 18 (large experts: 12) rows per expert
    assert moe.MOE_GROUP32_FROM_ROWS > moe.MOE_GROUP_MID_FROM_ROWS        # i.e. unused where the 64-row form exists
    assert moe.group16_row_limit(128, 2048, 768) == moe.MOE_GROUP16_ROWS_PER_EXPERT * 128
    assert moe.group16_row_limit(8, 4096, 14336) == moe.MOE_GROUP16_ROWS_PER_BIG_EXPERT * 8
