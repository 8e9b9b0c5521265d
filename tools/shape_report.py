"""Per-shape GEMM table of one bench pass: run `SVC_PROF_DUMP=gpurun_out/shapes.csv python bench.py ...` then
`python tools/shape_report.py gpurun_out/shapes.csv`.  Columns: class, M, N, K, epilogue, launches, total ms, TFLOP/s,
algorithmic GB/s -- sorted by total time."""
import sys

rows = []
for line in open(sys.argv[1]):
    c, M, N, K, epi, n, ms, fl, by = line.strip().split(",")
    rows.append((float(ms), int(c), int(M), int(N), int(K), int(epi), int(float(n)), float(fl), float(by)))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"total {tot:.1f} ms")
print(f"{'cls':>3} {'M':>8} {'N':>6} {'K':>6} {'epi':>3} {'n':>6} {'ms':>9} {'%':>5} {'us/launch':>9} {'TF':>7} {'GB/s':>7}")
for ms, c, M, N, K, epi, n, fl, by in rows[:40]:
    print(f"{c:>3} {M:>8} {N:>6} {K:>6} {epi:>3} {n:>6} {ms:>9.2f} {100 * ms / tot:>5.1f} {1e3 * ms / n:>9.1f} "
          f"{fl / ms / 1e9:>7.1f} {by / ms / 1e6:>7.1f}")
