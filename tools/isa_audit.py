#!/usr/bin/env python
"""ISA audit of the compositing kernels: compiles csrc/ggs_render.hip to gfx950 assembly (hipcc -S, no GPU needed), splits
every kernel into basic blocks and prices each block with the issue costs measured by tools/ubench/valu_rate.hip
(profiles/r02_valu_issue_rates.md): fma / mul / add class 2.8 cycles, other VALU 4.3, transcendental and permlane swaps
8.3 per wave-instruction.  Prints the blocks of the inner loops (the ones ending in a backward branch or sitting between
the loop labels) so the per-(tile, splat) and per-(tile, splat, quadrant) costs can be read off.
Usage: python tools/isa_audit.py [kernel-substring ...] > profiles/rNN_isa_audit.md"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gaussian-garments_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -mllvm " \
        "-amdgpu-atomic-optimizer-strategy=DPP -fno-gpu-rdc -fno-slp-vectorize".split()
CHEAP = re.compile(r"^v_(fma|fmac|mul|add|sub|subrev|mac)_f32")
SLOW = re.compile(r"^v_(exp|rcp|log|rsq|sqrt|sin|cos)_f32|^v_permlane(16|32)_swap")


def cost(op):
    if op.startswith("v_"):
        if SLOW.match(op):
            return "valu_slow", 8.3
        if CHEAP.match(op) and "dpp" not in op:
            return "valu_fma", 2.8
        return "valu_other", 4.3
    if op.startswith("s_nop"):
        return "nop", 0.0
    if op.startswith("s_"):
        return "salu", 0.0
    if op.startswith("ds_"):
        return "lds", 0.0
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem", 0.0
    return "other", 0.0


def kernels(asm):
    cur, body = None, []
    for ln in asm.splitlines():
        m = re.match(r"^(_Z\w+|ggs_k\w+):", ln)
        if m:
            cur, body = m.group(1), []
            continue
        if cur is not None:
            body.append(ln)
            if "s_endpgm" in ln and ".Lfunc_end" not in ln:
                pass
            if ln.startswith(".Lfunc_end"):
                yield cur, body
                cur = None


def blocks(body):
    out, label, ins = [], "entry", []
    for ln in body:
        t = ln.strip()
        if not t or t.startswith((";", ".")) and not re.match(r"^\.LBB\d+_\d+:", t):
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            if ins:
                out.append((label, ins))
            label, ins = m.group(1), []
            continue
        op = t.split()[0]
        ins.append((op, t))
    if ins:
        out.append((label, ins))
    return out


def main(filters):
    with tempfile.TemporaryDirectory() as td:
        s = os.path.join(td, "r.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-S", "--cuda-device-only", "-o", s, os.path.join(CSRC, "ggs_render.hip")],
                              stderr=subprocess.DEVNULL)
        asm = open(s).read()
    print("# ISA audit of ggs_render.hip (gfx950, hipcc -O3 -ffp-contract=off -fno-slp-vectorize)\n")
    print("Issue cost per wave64 instruction per SIMD (measured, profiles/r02_valu_issue_rates.md): `v_fma/mul/add_f32` 2.8 "
          "cycles, other VALU (cmp, cndmask, min, DPP, readlane, mov, cvt, integer) 4.3, `v_exp/v_rcp_f32` and "
          "`v_permlane*_swap` 8.3.  SALU / LDS / VMEM issue beside the VALU and are listed as counts only.\n")
    for name, body in kernels(asm):
        if filters and not any(f in name for f in filters):
            continue
        bl = blocks(body)
        tot = {}
        for _, ins in bl:
            for op, _ in ins:
                k, _c = cost(op)
                tot[k] = tot.get(k, 0) + 1
        print(f"## `{name}`\n\nstatic instruction counts: " + ", ".join(f"{k} {v}" for k, v in sorted(tot.items())) + "\n")
        print("| block | VALU fma-class | VALU other | VALU slow | VALU cycles | SALU | LDS | VMEM | s_nop | what |\n|---|---|---|---|---|---|---|---|---|---|")
        for label, ins in bl:
            c = {}
            cyc = 0.0
            for op, _ in ins:
                k, cc = cost(op)
                c[k] = c.get(k, 0) + 1
                cyc += cc
            nv = c.get("valu_fma", 0) + c.get("valu_other", 0) + c.get("valu_slow", 0)
            if nv < 6:
                continue
            ops = [op for op, _ in ins]
            what = []
            if any(o.startswith("v_exp") for o in ops):
                what.append("alpha test (exp2)")
            if any(o.startswith("v_rcp") for o in ops):
                what.append("backward quadrant body (rcp)")
            if any("permlane" in o for o in ops):
                what.append("gradient reduction")
            if any(o.startswith("global_atomic") for o in ops):
                what.append("atomic")
            if any(o.startswith("global_store") for o in ops):
                what.append("stores")
            if any(o.startswith("global_load") for o in ops):
                what.append("loads")
            print(f"| {label} | {c.get('valu_fma', 0)} | {c.get('valu_other', 0)} | {c.get('valu_slow', 0)} | {cyc:.0f} | "
                  f"{c.get('salu', 0)} | {c.get('lds', 0)} | {c.get('vmem', 0)} | {c.get('nop', 0)} | {', '.join(what)} |")
        print()


if __name__ == "__main__":
    main(sys.argv[1:])
