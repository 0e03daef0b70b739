"""Child process of test_gpu_train.py::test_data_parallel_machinery_on_one_rank (needs a GPU)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from action_conditioned_gans_amd import graph as G, optim, train as T   # noqa: E402
from oracle import models as OM   # noqa: E402
import train_cases as TC   # noqa: E402


def main():
    """argv[1] = 'default': the shipped data-parallel path (stream-ordered collectives, eager launches) and the exact-
    global-batch mode; 'experimental': the modes that capture HIP graphs around or with collectives (side-stream
    all-reduce between graph segments, ACG_CAPTURE_COLLECTIVES=1) - kept apart because a process that captures graphs
    while all-reduce work is outstanding aborted intermittently on the GPU box."""
    which = sys.argv[1] if len(sys.argv) > 1 else 'default'
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    x, y, a, s = TC.MG.inputs(2)
    finals = []
    modes = ((False, 'stream'), (True, 'stream')) if which == 'default' else ((False, 'stream'), (True, 'side'))
    for force, coll in modes:
        adv, loss, opt, dna, batch, ksize = TC.MG.CASES['c4_dna_wass_rmsprop']
        G.reset_default_graph()
        optim.set_data_parallel(1, force=force, collectives=coll)
        sess = G.Session(device='cuda:0')
        tr = T.Trainer(sess, True, 'bce', 'rmsprop', dna, batch_size=batch, ksize=ksize)
        sess.run(G.global_variables_initializer())
        params = OM.init_params(dna, batch=batch, ksize=ksize, seed=TC.MG.PARAM_SEED, dtype=torch.float32)
        for n, v in G.get_default_graph().variables.items():
            sess.set_value(v, params[n])
        for _ in range(4):                  # eager, capture, replay, replay
            tr.train_d(x, y, a)
            tr.train_g(x, y, a, s)
        torch.cuda.synchronize()
        finals.append({n: sess.get_value(v) for n, v in G.get_default_graph().variables.items()})
        if force:
            kinds = [type(o).__name__ for o in G.get_default_graph().ops]
            assert kinds.count('AllReduceOp') >= (4 if coll == 'side' else 2), kinds.count('AllReduceOp')
            if coll == 'side':     # host-side collectives between HIP-graph segments
                progs = [p for p in sess._programs.values() if any(k == 'host' for k, _ in p.segments)]
                assert progs and all(any(g is not None for g in p.graphs) for p in progs if p.runs >= 2)
            else:                  # stream-ordered collectives: the training programs are launched eagerly, one segment
                progs = [p for p in sess._programs.values() if p.eager]
                assert progs and all(p.graphs is None and len(p.segments) == 1 for p in progs), [(p.graphs, len(p.segments)) for p in progs]
    for n in finals[0]:
        assert torch.equal(finals[0][n], finals[1][n]), n
    adv, loss, opt, dna, batch, ksize = TC.MG.CASES['c4_dna_wass_rmsprop']
    params = OM.init_params(dna, batch=batch, ksize=ksize, seed=TC.MG.PARAM_SEED, dtype=torch.float32)
    if which == 'default':
        exact_mode(finals, x, y, a, s, params, dna, batch, ksize)
    else:
        captured_mode(finals, x, y, a, s, params, dna, batch, ksize)
    print('DP_ONE_RANK_OK', flush=True)
    os._exit(0)     # leave without communicator teardown


def exact_mode(finals, x, y, a, s, params, dna, batch, ksize):
    # synchronised BatchNorm on the one-rank communicator: global statistics = local ones, so the run must track the
    # plain one (different kernels: compared at 1e-4 of the weight scale after the same four steps)
    G.reset_default_graph()
    optim.set_data_parallel(1, force=True, exact_global_batch=True)    # SyncBN + GDL scale (x1) + global state-loss norm
    sess = G.Session(device='cuda:0')
    tr = T.Trainer(sess, True, 'bce', 'rmsprop', dna, batch_size=batch, ksize=ksize)
    sess.run(G.global_variables_initializer())
    for n, v in G.get_default_graph().variables.items():
        sess.set_value(v, params[n])
    for _ in range(4):
        tr.train_d(x, y, a)
        tr.train_g(x, y, a, s)
    torch.cuda.synchronize()
    kinds = [type(o).__name__ for o in G.get_default_graph().ops]
    assert kinds.count('BnMomentsAllReduceOp') >= 15 and kinds.count('BnSumsAllReduceOp') >= 15
    assert kinds.count('L2GlobalGradOp') >= 1 and kinds.count('ScalarAllReduceOp') >= 1
    for n, v in G.get_default_graph().variables.items():
        got, want = sess.get_value(v).double(), finals[0][n].double()
        assert (got - want).abs().max().item() <= 1e-4 * max(want.abs().max().item(), 1e-3) + 1e-6, n


def captured_mode(finals, x, y, a, s, params, dna, batch, ksize):
    # experimental: all-reduces captured into the HIP graph (ACG_CAPTURE_COLLECTIVES=1) - one graph per program again,
    # weights bit-identical to the plain run
    os.environ['ACG_CAPTURE_COLLECTIVES'] = '1'
    try:
        G.reset_default_graph()
        optim.set_data_parallel(1, force=True)
        sess = G.Session(device='cuda:0')
        tr = T.Trainer(sess, True, 'bce', 'rmsprop', dna, batch_size=batch, ksize=ksize)
        sess.run(G.global_variables_initializer())
        for n, v in G.get_default_graph().variables.items():
            sess.set_value(v, params[n])
        for _ in range(4):
            tr.train_d(x, y, a)
            tr.train_g(x, y, a, s)
        torch.cuda.synchronize()
        progs = [p for p in sess._programs.values() if p.runs >= 2 and p.graphs is not None]
        assert progs and all(len(p.segments) == 1 for p in progs), [len(p.segments) for p in progs]
        for n, v in G.get_default_graph().variables.items():
            assert torch.equal(sess.get_value(v), finals[0][n]), n
    finally:
        del os.environ['ACG_CAPTURE_COLLECTIVES']


if __name__ == '__main__':
    main()
