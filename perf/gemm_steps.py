"""Per-step cost of the lockstep skinny-GEMM kernel: one gate|up-shaped launch (2 x 14336 x 4096, tcomb_6_7) at batch n with the K
split forced by QPAL_GEMM_SK (one process per setting: the knob is read once).  python perf/gemm_steps.py <n>"""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2:
    import torch, qpalette_amd as qp
    n = int(sys.argv[1]); dev = torch.device("cuda", 0)
    q = "tcomb_6_7_0.5_none_0.9"
    copies = 8
    mods = [[qp.make_linear_from_info(q, qp.mem_op.dummy_linear_info(4096, 14336, q, seed=c * 2 + i, device=dev, codebook_seed=7)).to(dev) for i in range(2)] for c in range(copies)]
    qp.share_codebooks([m for pair in mods for m in pair])
    x = torch.randn(n, 4096, device=dev).half()
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        for pair in mods: qp.multi_gemv(pair, x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for pair in mods: qp.multi_gemv(pair, x)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(20): g.replay()
        e1.record(s); torch.cuda.synchronize()
    print(json.dumps({"sk": os.environ.get("QPAL_GEMM_SK"), "us_per_launch_incl_memset": e0.elapsed_time(e1) * 1e3 / (20 * copies)}))
else:
    n = sys.argv[1] if len(sys.argv) > 1 else "64"
    for sk in ("1", "2", "4", "8"):
        r = subprocess.run([sys.executable, __file__, n, "child"], env=dict(os.environ, QPAL_GEMM_SK=sk), capture_output=True, text=True)
        print("batch", n, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
