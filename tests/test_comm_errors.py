"""Error paths of the gradient transport (comm.py) on the CPU, against a stubbed librccl (tests/stub/stub_rccl.c):
a failing ncclCommInitRank / ncclAllReduce / ncclCommDestroy is a CommError carrying RCCL's message, the failing rank's
process ends with a non-zero exit code, and nobody hangs (every subprocess runs under a timeout).  With the real library a
rank that fails ncclCommInitRank leaves its peers inside theirs until RCCL's own timeout; what is pinned here is this
package's side: the error is raised, not swallowed, not retried, and not turned into a fallback."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope='module')
def stub(tmp_path_factory):
    out = str(tmp_path_factory.mktemp('stub') / 'libstub_rccl.so')
    subprocess.check_call(['gcc', '-shared', '-fPIC', '-O1', '-o', out, os.path.join(HERE, 'stub', 'stub_rccl.c')])
    return out


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _run_ranks(stub, mode, env_extra, world=2, timeout=120):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, OMP_NUM_THREADS='1', **env_extra)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'comm_error_worker.py'), str(r), str(world), str(port), stub, mode],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail('a rank hung (mode %s, %r)' % (mode, env_extra))
        outs.append((p.returncode, so, se))
    return outs


@pytest.mark.timeout(300)
def test_unique_id_reaches_every_rank_through_the_stub(stub):
    """Control: nothing fails - the stub's ncclCommInitRank accepts only the exact 128 bytes rank 0 generated (NULs from
    byte 9 on), so both ranks coming up means the bootstrap delivered them intact."""
    outs = _run_ranks(stub, 'allreduce', {})
    for rc, so, se in outs:
        assert rc == 0, se
        assert 'done' in so


@pytest.mark.timeout(300)
def test_comm_init_rank_failure_is_a_commerror_and_a_nonzero_exit(stub):
    outs = _run_ranks(stub, 'init', {'ACG_STUB_FAIL_INIT_RANK': '1'})
    (rc0, so0, se0), (rc1, so1, se1) = outs
    assert rc1 != 0 and 'CommError' in se1 and 'ncclCommInitRank failed' in se1 and 'unhandled system error' in se1
    assert 'communicator up' not in so1
    assert rc0 == 0 and 'done' in so0, se0          # the stub does not rendezvous: the healthy rank is not held up by THIS package


@pytest.mark.timeout(300)
def test_all_reduce_failure_is_a_commerror_and_a_nonzero_exit(stub):
    outs = _run_ranks(stub, 'allreduce', {'ACG_STUB_FAIL_ALLREDUCE': '2'})
    for rc, so, se in outs:
        assert rc != 0 and 'CommError' in se and 'ncclAllReduce failed' in se and 'invalid argument' in se
        assert 'communicator up' in so and 'done' not in so


def test_all_reduce_failure_surfaces_from_session_run(stub, monkeypatch):
    """Inside a training program the all-reduce is a launch-list op (optim.AllReduceOp -> all_reduce_ptr): its failure must
    come out of Session.run as the CommError it is - on the eager first run and on later runs alike - and must leave the
    optimizer update behind it unexecuted."""
    import train_cases as TC
    from oracle import cbind
    from action_conditioned_gans_amd import comm as C, graph as G
    monkeypatch.setenv('ACG_STUB_FAIL_ALLREDUCE', '3')           # 1, 2: the D step's and the G step's; 3: the second D step's
    lib = C.load_rccl(stub)
    comm = C.RcclCommunicator('cpu', 1, 0, lib=lib)
    sess, tr = TC.build_trainer(lambda **kw: G.Session(device='cpu', lib=cbind.load(), comm=comm, **kw), 'dna_k6_bce_rmsprop',
                                world_size=1, force_dp=True)
    rng = np.random.default_rng(3)
    x, y = (rng.uniform(-1, 1, (2, 64, 64, 3)).astype(np.float32) for _ in range(2))
    a, s = rng.standard_normal((2, 10)).astype(np.float32), rng.standard_normal((2, 5)).astype(np.float32)
    tr.train_d(x, y, a)
    tr.train_g(x, y, a, s)
    assert comm.calls == 2
    before = tr.d_opt_op.inputs[0].buf.clone()
    with pytest.raises(C.CommError, match='ncclAllReduce failed: invalid argument'):
        tr.train_d(x, y, a)
    assert torch.equal(before, tr.d_opt_op.inputs[0].buf), 'the optimizer ran behind a failed all-reduce'
    sess.close()


def test_destroy_failure_is_reported_once(stub, monkeypatch):
    from action_conditioned_gans_amd import comm as C
    monkeypatch.setenv('ACG_STUB_FAIL_DESTROY', '1')
    comm = C.RcclCommunicator('cpu', 1, 0, lib=C.load_rccl(stub))
    with pytest.raises(C.CommError, match='ncclCommDestroy failed: internal error'):
        comm.destroy()
    comm.destroy()          # the handle is gone: a second close() (Session.__exit__ after an explicit close) is a no-op


def test_a_short_unique_id_is_a_bootstrap_error():
    from action_conditioned_gans_amd import comm as C
    with pytest.raises(C.CommError, match='expected 128 bytes'):
        C.unpack_unique_id(b'\x2b\xad\xf0\x0d\xde\xad\xbe\xef\x02')
