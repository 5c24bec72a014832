import sys, numpy as np
a = np.load(sys.argv[1]); b = np.load(sys.argv[2])
bad = 0
for k in a.files:
    same = np.array_equal(a[k], b[k], equal_nan=True) if a[k].dtype.kind == 'f' else np.array_equal(a[k], b[k])
    print(k, a[k].shape, "bitwise equal" if same else "DIFFERENT")
    bad += not same
print("AB", "OK" if not bad else "FAILED")
