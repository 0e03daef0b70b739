"""One rank of tests/test_comm_errors.py: builds an RcclCommunicator over the stub librccl (tests/stub/stub_rccl.c) the way a
training process does - gloo control group for the unique-id bootstrap, then ncclCommInitRank - and lets a failure end the
process the way train.py would: an uncaught CommError, i.e. a traceback and exit code 1.
    python comm_error_worker.py <rank> <world> <port> <stub.so> <mode>     mode: init | allreduce"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import torch                           # noqa: E402
import torch.distributed as dist       # noqa: E402

from action_conditioned_gans_amd import comm as C   # noqa: E402


def main():
    rank, world, port, stub, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    comm = C.RcclCommunicator('cpu', world, rank, lib=C.load_rccl(stub))        # raises CommError on the failing rank
    print('rank %d: communicator up' % rank, flush=True)
    if mode == 'allreduce':
        t = torch.ones(8)
        for _ in range(3):
            comm.all_reduce(t)                                                  # the n-th call fails where the test says so
    comm.destroy()
    print('rank %d: done' % rank, flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
