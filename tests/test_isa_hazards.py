"""CPU-side check of the compiled gfx950 ISA (no GPU): the inline-asm MFMA loops and the wide epilogue stores keep the wait
states tools/check_mfma_hazards.py demands. The rules are exercised on hand-written snippets first, then on every kernel of
the library build (road_segmentation_unet_amd/csrc/build/*-gfx950.s, kept by the Makefile)."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_mfma_hazards as chk  # noqa: E402

HEAD = "\t.text\nk_test:\n"
MFMA = "\t;;#ASMSTART\n\tv_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[12:15], v[0:3]\n\t;;#ASMEND\n"


def _count(tmp_path, body):
    p = tmp_path / "k.s"
    p.write_text(HEAD + body + "\ts_endpgm\n")
    return chk.check(str(p))


def test_rule_results_read_too_early(tmp_path, capsys):
    assert _count(tmp_path, MFMA + "\ts_nop 3\n\tv_cvt_pk_bf16_f32 v20, v0, v1\n") == 1
    assert _count(tmp_path, MFMA + "\ts_nop 15\n\tv_cvt_pk_bf16_f32 v20, v0, v1\n") == 0
    # through a branch: the copy on the taken edge is as close as the fall-through one
    assert _count(tmp_path, MFMA + "\ts_cbranch_scc1 .LBB0_2\n\ts_nop 15\n.LBB0_2:\n\tv_mov_b32_e32 v20, v2\n") == 1
    # a builtin (compiler-managed) MFMA is not this script's business
    assert _count(tmp_path, "\tv_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[12:15], v[0:3]\n\tv_mov_b32_e32 v20, v2\n") == 0


def test_rule_operand_written_too_late(tmp_path, capsys):
    assert _count(tmp_path, "\tv_mov_b32_e32 v9, 0\n" + MFMA) == 1
    assert _count(tmp_path, "\tv_mov_b32_e32 v9, 0\n\ts_nop 3\n" + MFMA) == 0
    assert _count(tmp_path, "\tds_read_b128 v[8:11], v30\n" + MFMA) == 0  # LDS returns are ordered by s_waitcnt, not wait states


def test_rule_wide_store_data_overwritten(tmp_path, capsys):
    st = "\tbuffer_store_dwordx4 v[32:35], v139, s[40:43], s21 offen\n"
    assert _count(tmp_path, st + "\tv_add_u32_e32 v32, s99, v115\n") == 1  # the sequence that corrupted the convT outputs
    assert _count(tmp_path, st + "\ts_nop 3\n\tv_add_u32_e32 v32, s99, v115\n") == 0
    assert _count(tmp_path, st + "\tv_add_u32_e32 v36, s99, v115\n") == 0
    assert _count(tmp_path, "\tglobal_store_dwordx4 v[4:5], v[32:35], off\n\tv_mov_b32_e32 v33, 0\n") == 1
    assert _count(tmp_path, "\tbuffer_store_dwordx2 v[32:33], v139, s[40:43], s21 offen\n\tv_mov_b32_e32 v32, 0\n") == 0


def test_library_isa_is_clean():
    csrc = os.path.join(ROOT, "road_segmentation_unet_amd", "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "build", "*-gfx950.s")))
    if len(files) < 4:  # an older build tree without the saved ISA: rebuild (about half a minute)
        subprocess.check_call(["make", "-s", "-j4", "-B", "-C", csrc])
        files = sorted(glob.glob(os.path.join(csrc, "build", "*-gfx950.s")))
    names = [os.path.basename(f).split("-")[0] for f in files]
    for want in ("igemm_fwd2", "igemm_pp", "igemm_wgrad", "igemm_wgpp", "igemm_ct", "elementwise"):
        assert want in names, "no saved ISA for %s" % want
    bad = sum(chk.check(f) for f in files)
    assert bad == 0, "%d register hazards in the compiled kernels (see the captured output)" % bad


def test_rule_asm_vmem_reads_valu_written_sgpr(tmp_path, capsys):
    st = "\t;;#ASMSTART\n\tbuffer_store_dwordx4 v[32:35], v139, s[88:91], s21 offen\n\t;;#ASMEND\n"
    assert _count(tmp_path, "\tv_readlane_b32 s91, v228, 39\n" + st) == 1
    assert _count(tmp_path, "\tv_readlane_b32 s91, v228, 39\n\t;;#ASMSTART\n\ts_nop 4\n\tbuffer_store_dwordx4 v[32:35], v139, s[88:91], s21 offen\n\t;;#ASMEND\n") == 0
    assert _count(tmp_path, "\tv_readlane_b32 s80, v228, 39\n" + st) == 0
    # the compiler covers its own (non-asm) VMEM instructions
    assert _count(tmp_path, "\tv_readlane_b32 s91, v228, 39\n\tbuffer_store_dwordx2 v[32:33], v139, s[88:91], s21 offen\n") == 0


def test_rule_exchange_result_read_before_its_wait(tmp_path, capsys):
    bp = "\t;;#ASMSTART\n\tds_bpermute_b32 v10, v193, v10\n\t;;#ASMEND\n"
    bp2 = "\t;;#ASMSTART\n\tds_bpermute_b32 v11, v193, v11\n\t;;#ASMEND\n"
    wait = lambda n: "\t;;#ASMSTART\n\ts_waitcnt lgkmcnt(%d)\n\t;;#ASMEND\n" % n
    assert _count(tmp_path, bp + "\tv_mov_b32_e32 v20, v10\n" + wait(0)) == 1          # a copy scheduled in front of the wait
    assert _count(tmp_path, bp + wait(0) + "\tv_mov_b32_e32 v20, v10\n") == 0
    assert _count(tmp_path, bp + bp2 + wait(1) + "\tv_and_b32_e32 v20, v10, v9\n") == 0  # counted wait: one exchange may stay out
    assert _count(tmp_path, bp + bp2 + wait(1) + "\tv_and_b32_e32 v20, v11, v9\n") == 1  # ... but not the one that is read
    assert _count(tmp_path, bp + "\tv_mov_b32_e32 v20, v12\n" + wait(0)) == 0
