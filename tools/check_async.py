#!/usr/bin/env python3
"""Build-time guard for the asm-issued vector memory of the pencil sweep (csrc/stfem_pencil.hip).

The hot loads of that kernel are issued from asm statements the compiler does not track, and are
waited for with hand-counted s_waitcnt vmcnt(N).  Two things would break them silently:
  1. a spill: the compiler stores a load's destination register to scratch right after the asm
     statement, before the data has landed;
  2. any instruction touching a destination register while its load can still be in flight
     (a copy made by the register allocator, or a wait count that is too large).
This script disassembles the given objects and fails (exit 1) if an asynchronous-load kernel
(st_sweep_pencil<..., ADD = false, ...>) has VGPR spills / scratch, or if - scanning every such
kernel in program order - an instruction reads or writes a VGPR that is the destination of a
vector-memory load not yet covered by an s_waitcnt vmcnt (vmcnt retires in order: after
vmcnt(N) only the N youngest vector-memory instructions can be outstanding).  Loop bodies are
scanned twice, so a load carried over the back edge is seen too.

  usage: tools/check_async.py obj1.o [obj2.o ...]"""
import re
import subprocess
import sys
import tempfile
import os

LLVM = "/opt/rocm/lib/llvm/bin"


def unbundle(obj, tmp):
    fat = os.path.join(tmp, "fat.bin")
    elf = os.path.join(tmp, "k.elf")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={elf}"])
    return elf


def vregs(tok):
    out = []
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok):
        out += list(range(int(a), int(b) + 1))
    out += [int(a) for a in re.findall(r"\bv(\d+)\b", tok)]
    return out


def check_kernel(name, body):
    """body: list of (mnemonic, operand string).  Returns a list of violation strings."""
    bad = []
    vmem = []      # program-ordered outstanding vector-memory instructions: set of destination VGPRs (empty for stores)
    for rnd in range(2):  # second pass: state carried over loop back edges
        for n, (op, args) in enumerate(body):
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", args)
                if m:
                    keep = int(m.group(1))
                    vmem = vmem[len(vmem) - keep:] if keep else []
                continue
            inflight = set().union(*vmem) if vmem else set()
            ops = args.split(",")
            is_vmem = op.startswith(("global_", "buffer_", "flat_", "scratch_"))
            touched = set(vregs(args))
            if is_vmem and "load" in op:
                dst = set(vregs(ops[0]))
                hit = (touched - dst) & inflight  # address registers in flight
                # writing a destination that is still in flight (WAW) is reported too
                hit |= dst & inflight
                if hit:
                    bad.append(f"{name}: instr {n} {op} {args.strip()} touches in-flight v{sorted(hit)}")
                vmem.append(dst)
            else:
                hit = touched & inflight
                if hit:
                    bad.append(f"{name}: instr {n} {op} {args.strip()} touches in-flight v{sorted(hit)}")
                if is_vmem:
                    vmem.append(set())
            if len(vmem) > 64:
                vmem = vmem[-64:]
            if op == "s_endpgm":
                break
    return sorted(set(bad))


def main():
    failed = False
    for obj in sys.argv[1:]:
        with tempfile.TemporaryDirectory() as tmp:
            elf = unbundle(obj, tmp)
            notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", elf], text=True)
            dis = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", elf], text=True)
        # --- resource usage from the code-object metadata
        cur = {}
        kernels = {}
        for line in notes.splitlines():
            m = re.match(r"\s+\.(name|vgpr_spill_count|private_segment_fixed_size):\s+(\S+)", line)
            if m:
                cur[m.group(1)] = m.group(2)
            if line.strip().startswith(".wavefront_size") or line.strip().startswith("- .agpr_count"):
                pass
            if "name" in cur and "vgpr_spill_count" in cur and "private_segment_fixed_size" in cur:
                kernels[cur["name"]] = (int(cur["vgpr_spill_count"]), int(cur["private_segment_fixed_size"]))
                cur = {}
        # --- disassembly per function
        funcs = {}
        name = None
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                name = m.group(1)
                funcs[name] = []
                continue
            if name is None or not line.strip():
                continue
            t = line.split("//")[0].strip()
            if not t:
                continue
            parts = t.split(None, 1)
            funcs[name].append((parts[0], parts[1] if len(parts) > 1 else ""))
        checked = 0
        for kname, body in funcs.items():
            # st_sweep_pencil<P, NBM, TY, ADD, COEF>: the asynchronous instantiations have ADD = false ("Lb0E" first)
            # (fp64 only: the fp32 instantiations use compiler-tracked loads)
            m = re.search(r"3f64.*st_sweep_pencilILi\d+ELi\d+ELi\d+ELb([01])ELb[01]E", kname)
            if not m or m.group(1) != "0":
                continue
            checked += 1
            spills, scratch = kernels.get(kname, (None, None))
            if spills is None:
                print(f"{obj}: no metadata for {kname}")
                failed = True
            elif spills or scratch:
                print(f"{obj}: {kname}: {spills} spilled VGPRs, {scratch} B scratch - asynchronous loads are unsafe here")
                failed = True
            for b in check_kernel(kname, body)[:10]:
                print(f"{obj}: {b}")
                failed = True
        # tile kernels st_sweep_cart_tile<P, NBM, MINW, ADD, COEF, GEN, COLOUR>: ASYNC_LOADS = !GEN && NBM <= 3 (asm-issued
        # src prefetch, load_plane_async): no spills allowed either (no dataflow check: their other loads are the compiler's)
        for kname in funcs:
            m = re.search(r"st_sweep_cart_tileILi\d+ELi(\d+)ELi\d+ELb[01]ELb[01]ELb([01])ELi[01]E", kname)
            if not m or m.group(2) != "0" or int(m.group(1)) > 3:
                continue
            checked += 1
            spills, scratch = kernels.get(kname, (None, None))
            if spills is None:
                print(f"{obj}: no metadata for {kname}")
                failed = True
            elif spills or scratch:
                print(f"{obj}: {kname}: {spills} spilled VGPRs, {scratch} B scratch - asynchronous loads are unsafe here")
                failed = True
        print(f"{obj}: {checked} asynchronous-load kernels checked")
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
