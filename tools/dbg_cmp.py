import sys, numpy as np
a = np.load(sys.argv[1]); b = np.load(sys.argv[2])
for k in a.files:
    p, q = a[k], b[k]
    if p.shape != q.shape:
        print(k, "shape", p.shape, q.shape); continue
    m = np.argwhere(p.view(np.uint32) != q.view(np.uint32))
    if len(m):
        print(k, p.shape, "mismatches", len(m), "first", m[0].tolist(), "last", m[-1].tolist(),
              [(float(p[tuple(i)]), float(q[tuple(i)])) for i in m[:4]])
        rows = sorted(set(tuple(i[:-1]) for i in m.tolist()))
        print("   rows", rows[:8], "positions", sorted(set(i[-1] for i in m.tolist()))[:12])
print("compared", len(a.files))
