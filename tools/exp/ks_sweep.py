import sys, os
sys.path.insert(0, "/root/repo"); sys.argv=["x","none"]; __file__="/root/repo/tools/ab_gemm256.py"
exec(open("/root/repo/tools/ab_gemm256.py").read().split('which = sys.argv[1]')[0])
for M in (1536, 2048):
    for ks in (1, 2, 3, 4, 5, 7):
        run_splitk(M, 3072, 8192, ks)
    run_swiglu(M, 8192, 3072)
