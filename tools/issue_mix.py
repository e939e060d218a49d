#!/usr/bin/env python3
"""VALU opcode mix of the hot blocks of k_trace_scoped_persistent<12,false> and the issue ceiling that follows from it.

Builds trace.hip with -DPTC_MARKERS (comment lines at the start of the record step in both forms, the pushes, the leaf test
and the pop), reads the kernel out of `hipcc -S`, counts the VALU instructions of each region (a region runs from its marker to
the next one in program order) by ISSUE CLASS, and weights the regions by their executions per ray (counted by the CPU model of
the kernel, tests/cpp/scoped_model.cpp, on soup-1M primaries).  The class rates are what harness/ptissue measured on the
MI355X (profiles/r03_issue_roof.jsonl, best of 2 and 8 waves per SIMD):
  full   32-bit VOP1 / VOP2 encodings of mov, add, sub, mul, and, or, xor, shifts              ~2.5 cycles per wave64 instruction
  fma    v_fma_f32 / v_fmac_f32 (three register reads)                                        ~3.0 - 3.8
  half   min / max / compare / cndmask, every VOP3 / VOP3P encoding (min3, max3, or3, lshl_add, div_scale, div_fixup, pk_*) ~4.2
  trans  v_rcp_f32                                                                            ~8.1
CPU only.  usage: tools/issue_mix.py > profiles/issue_roof.json"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pathtracer_amd", "csrc")
FLAGS = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt "
         "-fno-gpu-flush-denormals-to-zero -fno-slp-vectorize -DPTC_MARKERS -S --cuda-device-only").split()
# executions per ray on soup-1M primaries (tests/cpp/scoped_model.cpp): records, leaf tests, pops
EXEC = {"record_sign": 33.1, "record_general": 0.0, "pushes": 33.1, "leaf": 6.85, "pop": 21.3, "loop_end": 0.0}

rates = {}
for line in open(os.path.join(ROOT, "profiles", "r03_issue_roof.jsonl")):
    if line.startswith("{"):
        j = json.loads(line)
        rates[j["instruction"]] = max(rates.get(j["instruction"], 0.0), j["wave_instr_per_s"])
CLASS_RATE = {
    "full": (rates["v_mul_f32"] + rates["v_add_f32"] + rates["v_sub_f32"] + rates["v_xor_b32"] + rates["v_and_b32"] + rates["v_mov_b32"] + rates["v_add_u32"]) / 7.0,
    "fma": (rates["v_fma_f32"] + rates["v_fma_f32 (two distinct VGPR sources)"]) / 2.0,
    "half": (rates["v_max_f32"] + rates["v_min_f32"] + rates["v_max3_f32"] + rates["v_min3_f32"] + rates["v_cndmask_b32_e64 (sgpr pair)"] + rates["v_cmp_ge_f32 vcc"]
             + rates["v_cmp_ge_f32_e64 sgpr pair"] + rates["v_or3_b32"] + rates["v_lshl_add_u32"] + rates["v_div_fixup_f32"] + rates["v_div_scale_f32"]
             + rates["v_pk_mul_f32"] + rates["v_pk_add_f32"] + rates["v_min_u32"]) / 14.0,
    "trans": rates["v_rcp_f32"],
}


def issue_class(op):
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
        return "trans"
    if op.startswith(("v_fma_f32", "v_fmac_f32")):
        return "fma"
    if op.startswith(("v_min", "v_max", "v_cmp", "v_cndmask", "v_pk_", "v_med3", "v_div_", "v_mbcnt", "v_readlane", "v_readfirstlane", "v_writelane")):
        return "half"
    if op.endswith("_e32"):
        return "full"
    return "half"            # everything else LLVM prints without _e32 is a VOP3 encoding (or3, lshl_add, add3, and_or, lshl_or, _e64 ...)


with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "trace.s")
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [os.path.join(CSRC, "trace.hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
m = re.search(r"^(_ZN12_GLOBAL__N_125k_trace_scoped_persistentILi12ELb0E\w*):[^\n]*\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M)
body = m.group(2)
regions, cur = collections.OrderedDict(), None
for line in body.splitlines():
    line = line.strip()
    mk = re.match(r";\s*PTC_MARK (\w+)", line)
    if mk:
        cur = mk.group(1)
        regions.setdefault(cur, collections.Counter())
        continue
    if cur is None or not line.startswith("v_"):
        continue
    regions[cur][issue_class(line.split()[0])] += 1
# the last region ("pop") runs to the end of the loop body: cut it at the loop's back edge (the first s_cbranch after which the
# text leaves the loop) - approximated by the instruction count up to the next s_endpgm being small anyway
total_time, total_instr, mix = 0.0, 0.0, {}
for name, counts in regions.items():
    n = EXEC.get(name, 0.0)
    mix[name] = {"valu_per_execution": dict(counts), "executions_per_ray": n}
    for c, k in counts.items():
        total_instr += k * n
        total_time += k * n / CLASS_RATE[c]
json.dump({
    "wave_instr_per_s_kernel_mix": total_instr / total_time,
    "class_rates_wave_instr_per_s": CLASS_RATE,
    "cycles_per_wave64_instruction_per_simd_at_2.4GHz": {c: 1024 * 2.4e9 / r for c, r in CLASS_RATE.items()},
    "valu_instructions_per_ray_in_hot_blocks": total_instr,
    "hot_block_mix": mix,
    "source": "tools/issue_mix.py: `hipcc -S -DPTC_MARKERS` of trace.hip at this commit (regions from marker to marker in program order), "
              "class rates from profiles/r03_issue_roof.jsonl (harness/ptissue on the MI355X: independent VALU chains on every CU, best of 2 and 8 "
              "waves per SIMD), executions per ray from tests/cpp/scoped_model.cpp on soup-1M primaries",
}, sys.stdout, indent=1)
print()
