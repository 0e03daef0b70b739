import os, torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR','127.0.0.1'); os.environ.setdefault('MASTER_PORT','29566')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda',0))
x=torch.ones(1<<20, device='cuda')
dist.all_reduce(x); torch.cuda.synchronize()
g=torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        y=x*2
        dist.all_reduce(y)
        z=y+1
    g.replay(); torch.cuda.synchronize()
    print('CAPTURE_OK', float(z[0]))
except Exception as e:
    print('CAPTURE_FAIL', type(e).__name__, str(e)[:300])
os._exit(0)
