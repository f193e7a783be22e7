"""Per array of two .npz files (tools/state_dump2.py): NaN-pattern differences, number and size of value differences, byte equality.
    python tools/state_cmp.py a.npz b.npz"""
import numpy as np, sys
a = np.load(sys.argv[1]); b = np.load(sys.argv[2])
for t in a.files:
    x, y = a[t].astype(np.float64).ravel(), b[t].astype(np.float64).ravel()
    nx, ny = np.isnan(x), np.isnan(y)
    ok = ~nx & ~ny
    d = np.abs(x[ok] - y[ok])
    print(t, "size", x.size, "nan pattern diff", int((nx != ny).sum()), "value diffs", int((d > 0).sum()), "max", d.max() if d.size else 0, "bytes equal", a[t].tobytes() == b[t].tobytes())
