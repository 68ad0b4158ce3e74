"""Build-time guard for the hand-issued gathers of the tap loops (tools/check_asm_gathers.py): no instruction may touch the
destination register of a `buffer_load_dword ... idxen` between the load and the s_waitcnt that retires it — the compiler cannot
know those registers are in flight, because the loads and their waits are inline asm.  Runs without a GPU: hipcc cross-compiles
the two translation units that contain such gathers and the checker walks every kernel's control-flow graph."""
import importlib.util
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_asm_gathers", os.path.join(ROOT, "tools", "check_asm_gathers.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)

KERNEL = """_Z1kv: ; @_Z1kv
	s_load_dwordx4 s[8:11], s[0:1], 0x0
	buffer_load_dword v10, v1, s[8:11], 0 idxen
	buffer_load_dword v11, v2, s[8:11], 0 idxen
	s_cbranch_scc1 .LBB0_2
.LBB0_1:
	v_add_f32_e32 v3, v4, v5
	s_branch .LBB0_3
.LBB0_2:
	%s
.LBB0_3:
	s_waitcnt vmcnt(1)
	v_cvt_f32_ubyte0 v6, v10
	%s
	s_waitcnt vmcnt(0)
	v_cvt_f32_ubyte0 v7, v11
	s_endpgm
	.size	_Z1kv, .Lfunc_end0-_Z1kv
"""


@pytest.mark.parametrize("in_branch,after_first_wait,violations", [
    ("v_mov_b32_e32 v8, v9", "v_mov_b32_e32 v8, v9", 0),
    ("v_mov_b32_e32 v10, v9", "v_mov_b32_e32 v8, v9", 1),          # a write to an in-flight destination on one branch only
    ("v_mov_b32_e32 v8, v9", "v_add_f32_e32 v8, v11, v9", 1),      # the second gather read after a wait that only retires the first
    ("v_mov_b32_e32 v8, v[10:11]", "v_mov_b32_e32 v8, v9", 2),     # a register range that covers both
])
def test_checker_on_synthetic_kernels(tmp_path, in_branch, after_first_wait, violations):
    p = tmp_path / "k.s"
    p.write_text(KERNEL % (in_branch, after_first_wait))
    bad, n = chk.check(str(p))
    assert n == 2 and len(bad) == violations, bad


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
@pytest.mark.parametrize("unit", ["pm_sweep.hip", "pm_sweep_lut.hip"])
def test_no_instruction_touches_a_gather_in_flight(tmp_path, unit):
    out = tmp_path / (unit + ".s")
    subprocess.run([os.path.join(ROOT, "tools", "isa.sh"), os.path.join(ROOT, "tsar-mvs_amd", "csrc", unit), str(out)], check=True, capture_output=True, timeout=900)
    bad, n = chk.check(str(out))
    assert n >= 36, "the translation unit no longer contains the asm-issued gathers this test guards"
    assert not bad, "\n".join(bad[:10])
    # Register budget of the production kernels, from the assembler's own metadata: every sweep kernel must fit the 128 VGPRs that four
    # waves per SIMD leave, WITHOUT scratch.  (Round 4 shipped, for an hour, a launch bound that read 1024 / BLK = 8 waves for the
    # 128-thread shape: 64 VGPRs + 224 B of scratch per lane — still bit-exact, so no parity test saw it, and cfg1 ran 22 % slower.)
    import re
    txt = out.read_text()
    seen = 0
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
        name, body = m.group(1), m.group(2)
        mv = re.search(r"pm_sweep_kernelILi(\d+)ELi\d+ELb[01]ELb1ELi(\d+)ELi(?:128|256)ELb([01])E", name)      # (last flag: the packed form)
        if not mv or int(mv.group(2)) == 0:          # (variant 0 = the generic one-tap loop of float imagery: not the bench path)
            continue
        seen += 1
        vgpr = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
        assert scratch == 0, f"{name}: {scratch} bytes of scratch per lane"
        if int(mv.group(1)) <= 4:                    # (the 32-entry best-N selection of long lists is allowed its extra registers)
            assert vgpr <= 128, f"{name}: {vgpr} VGPRs (four waves per SIMD need <= 128)"
    assert seen >= 4
    # the packed form of the box-11 kernels (pm_sweep_impl.h CMP) exists and keeps the same budget: it was written on a register diet
    # for exactly that (its first version held 141-150 VGPRs: three waves per SIMD, and lost 9 % before it gained anything)
    if unit == "pm_sweep.hip":
        assert len(re.findall(r"\.amdhsa_kernel \S*pm_sweep_kernelILi[24]ELi5ELb[01]ELb1ELi\d+ELi(?:128|256)ELb1E", txt)) >= 8


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_wmf_kernels_keep_their_keys_in_registers_at_two_waves_per_simd(tmp_path):
    """wmf_detect / wmf_fill (two lanes per pixel since round 5) hold 64 fp64 sort keys per lane in 128 VGPRs and run TWO waves per
    SIMD (__launch_bounds__(64, 2), wmf_kernels.hip).  If a compiler or flag change pushed the keys to scratch or AGPRs, or the LDS
    block past a seventh of a CU, the result would stay bit-exact and the kernels would be much slower — the failure this file
    exists for.  Recorded budget: 248 VGPRs, 0 AGPRs, 0 spills, 48 bytes of scratch per lane (the tap descriptor handed by reference
    to the one non-inlined sort function), 21 988 bytes of LDS (7 workgroups per CU), occupancy 2; 863 v_min_f64 (543 + 192
    compare-exchanges, 64 cross-stage minima, 64 NaN minima of the keys) and 735 v_max_f64: the network exists ONCE."""
    import re
    out = tmp_path / "wmf.s"
    subprocess.run([os.path.join(ROOT, "tools", "isa.sh"), os.path.join(ROOT, "tsar-mvs_amd", "csrc", "wmf_kernels.hip"), str(out)], check=True, capture_output=True, timeout=900)
    txt = out.read_text()
    seen = 0
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
        name, body = m.group(1), m.group(2)
        if "wmf_detect_kernel" not in name and "wmf_fill_kernel" not in name:
            continue
        seen += 1
        vgpr = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
        lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", body).group(1))
        assert scratch <= 64, f"{name}: {scratch} bytes of scratch per lane (recorded 48): sort keys or weights went to scratch"
        assert 128 <= vgpr <= 256, f"{name}: {vgpr} registers (two waves per SIMD need <= 256 including AGPRs)"
        assert lds <= 160 * 1024 // 7, f"{name}: {lds} bytes of LDS (seven workgroups per CU need <= 23 405)"
    assert seen == 2
    spills = [int(v) for v in re.findall(r"\.vgpr_spill_count:\s+(\d+)", txt)]
    assert spills and max(spills) == 0, spills
    agprs = [int(v) for v in re.findall(r"\.agpr_count:\s+(\d+)", txt)]
    assert agprs and max(agprs) == 0, agprs
    occ = [int(v) for v in re.findall(r"; Occupancy: (\d+)", txt)]
    assert occ and min(occ) >= 2, occ
    n_min, n_max = len(re.findall(r"v_min_f64", txt)), len(re.findall(r"v_max_f64", txt))
    assert n_max == 543 + 192 and 543 + 192 + 64 <= n_min <= 543 + 192 + 64 + 64 + 8, (n_min, n_max)
