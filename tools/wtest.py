import sys, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import cases
eng = cases.pkg.engine
p = np.random.default_rng(1).uniform(0,10,(768,1024)).astype(np.float32)
for d in ("/tmp", "/dev/shm"):
    w = eng.StackWriter(d + "/t.mha", 1024, 768, 20)
    t0 = time.perf_counter()
    for i in range(20): w.append(p)
    t1 = time.perf_counter(); w.finish(); t2 = time.perf_counter()
    print(d, "append ms", (t1-t0)/20*1e3, "finish ms", (t2-t1)*1e3)
    t0 = time.perf_counter(); open(d + "/raw.bin","wb").write(p.tobytes()*20); print(d, "raw write ms per plane", (time.perf_counter()-t0)/20*1e3)
