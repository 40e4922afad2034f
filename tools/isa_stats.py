#!/usr/bin/env python3
"""Static ISA digest of one kernel instantiation: the CPU-side proxy used while putting the persistent kernels on an
instruction diet (their point phase is VALU-issue-bound: 2 wavefronts per SIMD x instructions x 4 clocks).

    python tools/isa_stats.py slam-eds_amd/csrc/eds_fused.hip 'eds_fused6_kernelILi0ELi4ELi512ELi1ELi1E' [--flags "..."] [--dump out.s]

Compiles the source device-only to gfx950 assembly with the flags of csrc/Makefile, cuts out the named kernel, and prints for the
whole kernel and for its POINT PHASE (from the header of the solve loop to the first s_barrier inside it) the instruction count by
class, the packed-fp32 share, v_mov / v_cndmask / v_bfi counts, scratch traffic and the register budget."""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wno-unused-value -Wno-unused-function"
FUSED = "-mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-use-amdgpu-trackers=1"


def classify(op):
    if op.startswith("v_pk_"):
        return "valu_packed"
    if op.startswith(("v_mov_b32", "v_accvgpr")):
        return "valu_mov"
    if "dpp" in op:
        return "valu_dpp"
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
        return "valu_trans"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "valu_lane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def digest(lines, title):
    ops = collections.Counter()
    cls = collections.Counter()
    dpp = 0
    for ln in lines:
        s = ln.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        m = re.match(r"([a-z_0-9]+)", s)
        if not m:
            continue
        op = m.group(1)
        is_dpp = "quad_perm" in s or "row_" in s or "_dpp" in op
        ops[op] += 1
        cls[classify(op + ("_dpp" if is_dpp and not op.endswith("dpp") else ""))] += 1
    n = sum(ops.values())
    print(f"== {title}: {n} instructions")
    for k, v in sorted(cls.items(), key=lambda kv: -kv[1]):
        print(f"   {k:12s} {v:6d}  {100.0 * v / max(n, 1):5.1f} %")
    key = ["v_mov_b32", "v_cndmask_b32", "v_bfi_b32", "v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_fmac_f32",
           "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "ds_read_b128", "ds_write_b128", "global_load_dwordx4",
           "scratch_load_dword", "scratch_store_dword", "v_readlane_b32", "v_writelane_b32"]
    shown = {k: ops[k] for k in key if ops[k]}
    # e32/e64 spellings
    merged = collections.Counter()
    for k, v in ops.items():
        merged[re.sub(r"_(e32|e64|dpp|sdwa)$", "", k)] += v
    print("   " + "  ".join(f"{k}={merged[k]}" for k in key if merged[k]))
    scal = sum(v for k, v in merged.items() if re.match(r"v_(fma|mul|add|sub|fmac|mac)_f32$", k))
    pk = sum(v for k, v in merged.items() if re.match(r"v_pk_(fma|mul|add)_f32$", k))
    print(f"   fp32 arithmetic: {scal} scalar, {pk} packed")
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("kernel", help="substring of the mangled kernel name")
    ap.add_argument("--flags", default=None, help="extra compiler flags (default: the Makefile's for eds_fused.hip)")
    ap.add_argument("--dump", default=None)
    ap.add_argument("--asm", default=None, help="use this assembly file instead of compiling")
    a = ap.parse_args()
    if a.asm:
        text = open(a.asm).read()
    else:
        extra = a.flags if a.flags is not None else (FUSED if "eds_fused.hip" in a.src else "")
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "k.s")
            cmd = f"/opt/rocm/bin/hipcc {BASE} {extra} --offload-device-only -S {a.src} -o {out}"
            r = subprocess.run(cmd, shell=True, cwd=ROOT, capture_output=True, text=True)
            if r.returncode != 0:
                sys.stderr.write(r.stderr[-4000:])
                sys.exit(1)
            text = open(out).read()
    lines = text.splitlines()
    start = next((i for i, l in enumerate(lines) if re.match(r"_Z\w*:", l) and a.kernel in l), None)
    if start is None:
        sys.exit(f"kernel {a.kernel} not found")
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    body = lines[start:end]
    if a.dump:
        open(a.dump, "w").write("\n".join(body) + "\n")
    name = lines[start].split(":")[0]
    digest(body, name)
    # the point phase: the Depth=1 loop header with the longest run up to the next s_barrier
    best = None
    for i, l in enumerate(body):
        if "Loop Header: Depth=1" in l:
            j = next((k for k in range(i, len(body)) if body[k].strip().startswith("s_barrier")), None)
            if j is not None and (best is None or j - i > best[1] - best[0]):
                best = (i, j)
    if best:
        digest(body[best[0]:best[1]], "point phase (solve-loop header .. first s_barrier)")
    # register budget from the kernel descriptor comments
    tail = "\n".join(lines[end:end + 80])
    for key in ("NumVgprs", "NumAgprs", "TotalNumVgprs", "NumSgprs", "ScratchSize", "Occupancy", "LDSByteSize", "codeLenInByte"):
        m = re.search(rf"; {key}: (\d+)", tail)
        if m:
            print(f"   {key}: {m.group(1)}")
    m = re.search(r"; SGPRSpill (\d+)", tail) or re.search(r"sgpr_spill_count:\s+(\d+)", text[text.find(name, text.find('.amdhsa_kernel')):][:20000] if False else "")
    for key in ("sgpr_spill_count", "vgpr_spill_count"):
        mm = re.search(rf"\.name:\s+{re.escape(name)}\b.*?{key}:\s+(\d+)", text, re.S)
        mm2 = re.search(rf"{key}:\s+(\d+)(?:(?!\.name:).)*?\.name:\s+{re.escape(name)}\b", text, re.S)
        got = mm or mm2
        if got:
            print(f"   {key}: {got.group(1)}")


if __name__ == "__main__":
    main()
