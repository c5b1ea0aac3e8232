import ctypes, os, sys, subprocess
sys.path.insert(0, os.getcwd())
os.environ["MUGIQ_HIP_LIB"] = os.path.join(os.getcwd(), "tools/probes/build/libmugiq_hip_cm4.so")
sys.argv = ["bench_mg.py"]
exec(open("tools/bench_mg.py").read())
lib = ctypes.CDLL(os.environ["MUGIQ_HIP_LIB"])
buf = (ctypes.c_longlong * 64)()
print("rc", lib.mugiq_hip_debug_cm_stamps(buf))
for r in range(4):
    s = [buf[r * 8 + i] for i in range(8)]
    print("round", 4 + r, "vmcnt0", s[7] - s[6], "topBarrier", s[0] - s[7], "firstLoads", s[1] - s[0], "unit0", s[2] - s[1], "units1-5", s[3] - s[2], "shuffle+red", s[4] - s[3], "redBarrier", s[5] - s[4], "total", (buf[(r + 1) * 8 + 6] - s[6]) if r < 3 else None)
