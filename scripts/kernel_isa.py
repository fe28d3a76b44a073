"""Development aid: disassembly and resource use of one kernel of libsdft_hip.so's translation units.

    python scripts/kernel_isa.py f32f32 "forward_rows_kernel<float, 2, 3, false, 2, 0" [--dump out.s]

Prints registers / LDS / scratch from the code object's metadata and an instruction histogram (whole kernel and the
largest loop body by backward branch), no GPU needed.
"""
import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(combo, td):
    obj = os.path.join(ROOT, "sdft_amd", "lib", "obj", f"sdft_capi_{combo}.o")
    local = os.path.join(td, "dev.o")
    shutil.copy(obj, local)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], capture_output=True, text=True, cwd=td)
    cos = [f for f in os.listdir(td) if "amdgcn" in f and "gfx950" in f]
    return os.path.join(td, cos[0])


def main():
    combo, pat = sys.argv[1], sys.argv[2]
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    with tempfile.TemporaryDirectory() as td:
        co = code_object(combo, td)
        dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True).stdout
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    kernels, name = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            name = m.group(1); kernels[name] = []
        elif name and line.strip():
            kernels[name].append(line)
    mangled = list(kernels)
    plain = subprocess.run(["c++filt"], input="\n".join(mangled), capture_output=True, text=True).stdout.splitlines()
    hits = [(k, d) for k, d in zip(mangled, plain) if pat in d]
    for k, d in hits:
        body = kernels[k]
        print("==", d[:200])
        i = notes.find(f".name:           {k}\n")
        blk = notes[max(0, notes.rfind("  - .agpr_count", 0, i)):notes.find("  - .agpr_count", i) if notes.find("  - .agpr_count", i) > 0 else len(notes)]
        for key in (".vgpr_count", ".agpr_count", ".sgpr_count", ".group_segment_fixed_size", ".private_segment_fixed_size", ".vgpr_spill_count"):
            m = re.search(re.escape(key) + r":\s+(\d+)", blk)
            if m:
                print(f"   {key[1:]:28s} {m.group(1)}")
        ops = [l.split("//")[0].split()[0] for l in body if l.split("//")[0].strip()]
        print("   instructions:", len(ops))
        # largest loop: a backward branch to a label; count the instructions between target and branch
        addr = {}
        for idx, l in enumerate(body):
            m = re.search(r"//\s*([0-9A-Fa-f]+):", l)
            if m:
                addr[int(m.group(1), 16)] = idx
        best = (0, 0, 0)
        for idx, l in enumerate(body):
            m = re.match(r"\s*s_cbranch_\w+\s+(\d+)", l.split("//")[0])
            a = re.search(r"//\s*([0-9A-Fa-f]+):", l)
            t = re.search(r"<.*\+0x([0-9a-fA-F]+)>", l)
            if m and a and t:
                # objdump prints the target as <kernel+0xOFF>
                base = min(addr)
                tgt = base + int(t.group(1), 16)
                if tgt in addr and addr[tgt] < idx and idx - addr[tgt] > best[0]:
                    best = (idx - addr[tgt], addr[tgt], idx)
        if best[0]:
            loop = ops[best[1]:best[2] + 1]
            hist = collections.Counter(re.sub(r"_e(32|64)$", "", o) for o in loop)
            print(f"   largest loop: {len(loop)} instructions")
            fam = collections.Counter()
            for o, c in hist.items():
                f = "valu" if o.startswith("v_") else "salu" if o.startswith("s_") else "lds" if o.startswith("ds_") else "vmem" if o.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
                fam[f] += c
            print("   by unit:", dict(fam))
            print("   top:", ", ".join(f"{o} {c}" for o, c in hist.most_common(24)))
        if dump:
            with open(dump, "w") as fh:
                fh.write("\n".join(body))
    if not hits:
        print("no kernel matches; candidates:")
        for d in plain:
            if pat.split("<")[0] in d:
                print("  ", d[:160])


if __name__ == "__main__":
    main()
