"""Cost of the split-K hand-off per layer: separate reduction + one-launch BatchNorm against the BatchNorm kernel summing
the slabs itself (row layout / quad layout), 20 launches in a HIP graph.  python tools/handoff_probe.py [f32|bf16]"""
import ctypes, sys
import torch
sys.path.insert(0, '.')
from action_conditioned_gans_amd import _lib as L

lib = L.get()
dev = torch.device('cuda:0')
half = len(sys.argv) > 1 and sys.argv[1] == 'bf16'
dt = L.ACG_BF16 if half else L.ACG_F32
tdt = torch.bfloat16 if half else torch.float32
p = lambda t: ctypes.c_void_p(t.data_ptr())


def timed(fn, reps=20, rounds=5):
    st = torch.cuda.Stream(dev)
    with torch.cuda.stream(st):
        s = ctypes.c_void_p(st.cuda_stream)
        fn(s); st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps):
                fn(s)
        best = 1e9
        for _ in range(rounds):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st); g.replay(); b.record(st); st.synchronize()
            best = min(best, a.elapsed_time(b) * 1e3 / reps)
    return best


for rows, c, splits, groups in [(2048, 128, 4, 1), (2048, 128, 4, 2), (512, 256, 8, 1), (1024, 256, 4, 2), (128, 512, 16, 1), (256, 512, 8, 2), (2048, 64, 2, 1)]:
    slabs = torch.randn(splits, rows * c, device=dev)
    x = torch.zeros(rows, c, dtype=tdt, device=dev)
    y = torch.zeros(rows, c, dtype=tdt, device=dev)
    xr = torch.zeros(rows, c, device=dev)
    beta = torch.zeros(c, device=dev)
    mean, rstd = torch.zeros(groups * c, device=dev), torch.zeros(groups * c, device=dev)
    n = lib.bn_workspace_bytes(rows, c, groups)
    ws = torch.zeros(max(n, 16), dtype=torch.uint8, device=dev)
    rl = L.ReduceList()
    rl.slabs[0], rl.out[0], rl.numel[0], rl.splits[0], rl.accumulate[0] = slabs.data_ptr(), xr.data_ptr(), xr.numel(), splits, 0.0
    t_red = timed(lambda s: lib.splitk_reduce_many(ctypes.byref(rl), 1, s))
    t_bn = timed(lambda s: lib.bn_act_fwd(p(x), p(beta), p(y), p(mean), p(rstd), rows, c, c, c, groups, 1e-3, L.ACT_RELU, 0.2, dt, 0, p(ws), n, s))
    t_both = timed(lambda s: (lib.splitk_reduce_many(ctypes.byref(rl), 1, s), lib.bn_act_fwd(p(x), p(beta), p(y), p(mean), p(rstd), rows, c, c, c, groups, 1e-3, L.ACT_RELU, 0.2, dt, 0, p(ws), n, s)))
    res = []
    for layout in (0, 1):
        res.append(timed(lambda s: lib.bn_act_fwd_slabs(p(slabs), splits, p(x), p(beta), p(y), p(mean), p(rstd), rows, c, c, c, groups, 1e-3, L.ACT_RELU, 0.2, dt, layout, 0, p(ws), n, s)))
    print('%s rows %5d c %4d splits %2d groups %d: reduce %5.2f us  bn %5.2f us  reduce+bn %5.2f us | bn summing slabs: rows layout %5.2f us  quads %5.2f us'
          % ('bf16' if half else 'f32', rows, c, splits, groups, t_red, t_bn, t_both, res[0], res[1]), flush=True)
