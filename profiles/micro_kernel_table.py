"""rocprofv3 kernel-trace CSV of a conv_micro.py run -> per (kernel symbol, grid) rows: launches, average / minimum microseconds.
    python profiles/micro_kernel_table.py gpurun_out/<dir> [substring ...]"""
import csv
import glob
import os
import sys


def short(name):
    s = name[5:] if name.startswith("void ") else name
    s = s.replace("(anonymous namespace)::", "")
    depth = 0
    for i, ch in enumerate(s):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return s[:i]
    return s


def main():
    d, pats = sys.argv[1], sys.argv[2:]
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = {}
    order = []
    for r in csv.DictReader(open(f, newline="")):
        k = (short(r["Kernel_Name"]), r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))
        if pats and not any(p in k[0] for p in pats):
            continue
        if k not in rows:
            rows[k] = []
            order.append(k)
        rows[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k in order:
        v = rows[k]
        print(f"{k[0][:70]:70s} grid {int(k[1]) // 256 if k[1] else 0:6d} lds {k[2]:>7s} n {len(v):4d} avg {sum(v) / len(v):8.1f} min {min(v):8.1f}")


if __name__ == "__main__":
    main()
