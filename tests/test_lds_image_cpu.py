"""The LDS image of the attention kernels (simple-multimodal_amd/csrc/attn_helpers.h img_off) checked on the CPU: the
LDS-DMA fill map is the inverse of img_off, and both kinds of read are bank-conflict-free under the MI355X lane-group
rules (ds_read_b128: 4 groups of 16 lanes, ds_read_b64_tr_b16: 2 halves of 32; bank = (byte / 4) mod 64)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import lds_image_check as chk


def test_new_image_is_conflict_free_and_old_was_not():
    for DH in (96, 64):
        row_new, tr_new = chk.read_cycles(DH, chk.off_new)
        row_old, tr_old = chk.read_cycles(DH, chk.off_old)
        assert row_new == 4.0 and tr_new == 2.0          # the floor of each instruction
        assert row_old == 4.0 and tr_old == 4.0          # rounds 1-2: every transposed read 2-way conflicted


def test_dma_fill_map_inverts_img_off():
    for DH in (96, 64):
        assert chk.dma_map_ok(DH)


def test_tr_lane_parts_match_img_off():
    for DH in (96, 64):
        assert chk.tr_parts_ok(DH)


def test_gemm6_images_reads_and_piece_maps():
    """gemm6.hip uses the same image for its GEMM operand tiles: [256][BK] row-read tiles, [BK][256] transposed-read tiles in the
    standard MFMA k order; the lane parts / immediates of the kernel reproduce img_off, both reads sit on their conflict-free
    floor, and the 1-KiB LDS-DMA pieces cover each tile exactly once at the image's addresses."""
    for BK in (32, 64):
        assert chk.gemm6_row_read_cycles(BK) == 4
        ok, cyc = chk.gemm6_tr_read_ok(BK // 16)
        assert ok and cyc == 2
        assert chk.gemm6_piece_map_ok(False, BK) and chk.gemm6_piece_map_ok(True, BK)
