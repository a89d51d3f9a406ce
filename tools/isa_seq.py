"""Load / wait sequence of compiled kernels: where the compiler put the waits.

    hipcc --cuda-device-only -S -O3 --offload-arch=gfx950 -std=c++17 -Iinclude gnn_amd/csrc/rwk_spmm.hip -o /tmp/rwk_spmm.s
    python tools/isa_seq.py /tmp/rwk_spmm.s spmm_tile_kernel        # kernels whose mangled name contains the pattern
    python tools/isa_seq.py --scan /tmp/*.s                         # kernels with three or more loads waited one by one

L = global load, S = global store, wN = s_waitcnt vmcnt(N), | = workgroup barrier, . = branch.  "L w0 L w0 L w0" is a
chain of trips to memory; "L L L L w3 w2 w1 w0" is one trip."""
import re
import sys


def kernels(path):
    lines = open(path).read().splitlines()
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):\s", lines[i] + " ")
        if not m:
            i += 1
            continue
        seq = []
        i += 1
        while i < len(lines) and not lines[i].strip().startswith("s_endpgm"):
            line = lines[i].strip()
            if line.startswith(("global_load", "flat_load", "buffer_load")):
                seq.append("L")
            elif line.startswith(("global_store", "flat_store", "buffer_store")):
                seq.append("S")
            elif line.startswith("s_barrier"):
                seq.append("|")
            elif line.startswith("s_waitcnt") and "vmcnt" in line:
                seq.append("w" + re.search(r"vmcnt\((\d+)\)", line).group(1))
            elif line.startswith("s_cbranch"):
                seq.append(".")
            i += 1
        yield m.group(1), re.sub(r"(?:\. )+", ". ", " ".join(seq))
        i += 1


def main():
    if sys.argv[1] == "--scan":
        for path in sys.argv[2:]:
            for name, seq in kernels(path):
                flat = seq.replace(". ", "").replace("| ", "") + " "
                runs = re.findall(r"(?:L w0 ){3,}", flat)
                if runs:
                    print(path.split("/")[-1], name[:110], "runs:", len(runs), "longest:", max(len(r) // 5 for r in runs))
        return
    path, pat = sys.argv[1], sys.argv[2]
    for name, seq in kernels(path):
        if pat in name:
            print(name[:110])
            print("  ", seq[:3000])
            print()


if __name__ == "__main__":
    main()
