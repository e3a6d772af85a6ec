"""Per-kernel launch statistics out of a rocprofv3 (rocpd) database: python tools/rocpd_kernels.py <results.db> [substring]
Launches are grouped by (kernel, grid, workgroup) in order of first appearance; the first `skip` launches of a group
(warm-up) are left out of the average."""
import collections
import sqlite3
import sys


def main(path, sub="", skip=3):
    c = sqlite3.connect(path)
    agg = collections.OrderedDict()
    for n, s, e, g, w in c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"):
        if sub in n:
            agg.setdefault((n.split("(")[0][:64], g, w), []).append((e - s) / 1e3)
    print(f"{'kernel':64s} {'grid':>9s} {'wg':>4s} {'n':>4s} {'avg us':>9s} {'min us':>9s}")
    for k, v in agg.items():
        v = v[skip:] if len(v) > skip else v
        print(f"{k[0]:64s} {k[1]:9d} {k[2]:4d} {len(v):4d} {sum(v) / len(v):9.1f} {min(v):9.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
