import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
for k in a.files:
    d = np.argwhere(a[k] != b[k])
    print(k, "differing words:", len(d))
    for idx in d[:20]:
        print("  ", tuple(idx), int(a[k][tuple(idx)]), int(b[k][tuple(idx)]))
