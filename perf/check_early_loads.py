#!/usr/bin/env python3
"""Post-build guard for the hand-placed memory waits of the fused GEMV kernels.

csrc/tc_kernels.h issues its early-staging loads (x chunks, codebook-image entries, the rotation's inputs) from inline asm at a
wave's first instructions — deliberately outside the compiler's wait-count bookkeeping — and waits for them with ONE hand-written
`s_waitcnt vmcnt(0)` in front of the first weight loads.  Nothing in the language stops the register allocator from copying,
reusing or spilling a destination VGPR of such a load between its issue and that wait: the build would then corrupt data
silently.  This script disassembles every fused-GEMV code object of a build and checks, kernel by kernel:

  * every vector-memory load with an SGPR base in front of the kernel's job fetch (its first scalar load at a run-time offset:
    the inline-asm loads are issued from preloaded arguments before anything else happens) is followed by a
    `s_waitcnt vmcnt(N)` that covers it (vector-memory operations complete in issue order: N <= the number issued after it)
    on EVERY control-flow path BEFORE any instruction names one of its destination registers — as a source, as a destination,
    or as the data of a scratch / buffer store (a spill);
  * kernels that use scratch memory at all (`.private_segment_fixed_size` > 0) AND have such early loads are listed: the
    check above covers their spill code as well, and tests/test_capi_and_host.py pins the list, so that a new spilling
    instantiation is a decision, not an accident.

    python perf/check_early_loads.py [build dir]     -> exit code 1 and one line per violation
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
UNITS = ("tcq_gemv", "lut_gemv")  # object files whose kernels use early staging: tcq_gemv*.o, lut_gemv*.o

_REG = re.compile(r"\b([vas])(\d+)\b|\b([vas])\[(\d+):(\d+)\]")


def regs(text, kind="v"):
    """set of register numbers of class `kind` named in an operand string"""
    out = set()
    for m in _REG.finditer(text):
        if m.group(1):
            if m.group(1) == kind:
                out.add(int(m.group(2)))
        elif m.group(3) == kind:
            out.update(range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def kernels_of(code_object):
    """-> {kernel name: [(mnemonic, operand text)]} in program order, {kernel name: scratch bytes}"""
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", code_object], capture_output=True, text=True,
                         check=True).stdout
    kernels, cur, base = {}, None, 0
    for line in dis.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
        if m:
            cur = kernels.setdefault(m.group(2), [])
            base = int(m.group(1), 16)
            continue
        if cur is None or not line.startswith("\t"):
            continue
        text, _, comment = line.partition("//")
        text = text.strip()
        if not text:
            continue
        parts = text.split(None, 1)
        am = re.match(r"\s*([0-9A-Fa-f]+):", comment)
        tm = re.search(r"<.+\+0x([0-9a-f]+)>\s*$", comment)
        # (mnemonic, operands, address, branch target address or None)
        cur.append((parts[0], parts[1] if len(parts) > 1 else "", int(am.group(1), 16) if am else -1,
                    base + int(tm.group(1), 16) if tm and parts[0].startswith(("s_branch", "s_cbranch")) else None))
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", code_object], capture_output=True, text=True, check=True).stdout
    scratch, name = {}, None
    for line in notes.splitlines():
        m = re.search(r"\.name:\s+(\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"\.private_segment_fixed_size:\s+(\d+)", line)
        if m:
            pending = int(m.group(1))
            scratch["__pending__"] = pending
        m = re.search(r"\.symbol:\s+(\S+)\.kd", line)
        if m and "__pending__" in scratch:
            scratch[m.group(1)] = scratch.pop("__pending__")
    scratch.pop("__pending__", None)
    return kernels, scratch


def check_kernel(name, insts):
    """-> (number of early loads, [violation strings]).  Walks the control-flow graph from every early load: along EVERY path the
    load must be covered by a wait before one of its destination registers is named."""
    index_of = {a: i for i, (_, _, a, _) in enumerate(insts)}
    # The early region: everything in front of the job fetch — the first scalar load of the kernel-argument block at a RUN-TIME
    # offset (`s_load_dwordx8 s[..], s[..], s9 offset:0x78`: mp.job[j]).  The inline-asm loads are all issued there, from preloaded
    # arguments; the compiler's own saddr-form loads (the on-demand staging paths) come after it and carry the compiler's waits.
    job_fetch = next((i for i, it in enumerate(insts) if it[0].startswith("s_load_") and re.search(r"\],\s*s\d+\b", it[1])), len(insts))
    first_barrier = next((i for i, it in enumerate(insts) if it[0] == "s_barrier"), len(insts))
    early, bad = 0, []
    for i, (op, args, _, _) in enumerate(insts[:min(first_barrier, job_fetch)]):
        if not (op.startswith("global_load_") and not op.startswith("global_load_lds")):
            continue
        ops = [a.strip() for a in args.split(",")]
        if len(ops) < 3 or not ops[2].startswith("s["):  # (saddr form only: what the inline asm emits)
            continue
        early += 1
        dst = regs(ops[0])
        # state = (instruction index, vector-memory operations issued after the load so far): `s_waitcnt vmcnt(N)` has waited for
        # the load once N <= that count (vector-memory operations complete in issue order)
        todo, seen, found = [(i + 1, 0)], set(), None
        while todo and found is None:
            j, younger = todo.pop()
            while j < len(insts):
                if (j, younger) in seen:
                    break
                seen.add((j, younger))
                op2, args2, _, target = insts[j]
                if op2 == "s_waitcnt":
                    m = re.search(r"vmcnt\((\d+)\)", args2)
                    if m and int(m.group(1)) <= younger:
                        break
                elif op2 == "s_endpgm":
                    break  # (a wave may end with the load in flight: nothing reads the register)
                else:
                    touched = regs(args2) & dst
                    if touched:
                        found = (f"{name}: `{op2} {args2}` (instruction {j}) names v{sorted(touched)} of the early load `{op} {args}` "
                                 f"(instruction {i}) on a path with no covering s_waitcnt vmcnt in between")
                        break
                    if op2.startswith(("global_", "buffer_", "scratch_", "flat_")):
                        younger += 1
                    if target is not None and target in index_of:
                        if op2 == "s_cbranch_execz" and j + 1 < len(insts) and insts[j + 1][0] == "s_branch":
                            j += 1  # LLVM's "no lane live" companion of a uniform branch: every lane is live in these prologues
                            continue
                        if op2 == "s_branch":
                            j = index_of[target]
                            continue
                        todo.append((index_of[target], younger))
                j += 1
        if found:
            bad.append(found)
    return early, bad


def run(build_dir):
    objs = sorted(o for u in UNITS for o in glob.glob(os.path.join(build_dir, u + "*.o")))
    if not objs:
        raise SystemExit(f"no {UNITS} objects under {build_dir}: build the library first (make -C q-palette_amd/csrc)")
    violations, with_scratch, nk, nearly = [], [], 0, 0
    tmp = tempfile.mkdtemp(prefix="qpal_isa_")
    try:
        for obj in objs:
            local = os.path.join(tmp, os.path.basename(obj))
            shutil.copy(obj, local)
            subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], capture_output=True, check=True)
            for co in glob.glob(local + ".*gfx950"):
                kernels, scratch = kernels_of(co)
                for name, insts in kernels.items():
                    if "tc_gemv_kernel" not in name:
                        continue
                    nk += 1
                    early, bad = check_kernel(name, insts)
                    nearly += early
                    violations += bad
                    if early and scratch.get(name, 0) > 0:
                        with_scratch.append((os.path.basename(obj), name, scratch[name]))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {"kernels": nk, "early_loads": nearly, "violations": violations, "early_with_scratch": with_scratch}


def main():
    build_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "q-palette_amd", "csrc", "build")
    r = run(build_dir)
    print(f"{r['kernels']} fused-GEMV kernels, {r['early_loads']} early loads checked, {len(r['violations'])} violations")
    for obj, name, nbytes in r["early_with_scratch"]:
        print(f"  early loads + {nbytes} B of scratch: {obj}: {name}")
    for v in r["violations"]:
        print("VIOLATION", v)
    return 1 if r["violations"] else 0


if __name__ == "__main__":
    sys.exit(main())
