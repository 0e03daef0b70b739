"""Fixed cost of a bf16 conv launch: forward of 1x1 / 3x3 / 5x5 layers with 1 - 25 K-steps on a 32768 x 128 output (256 tiles of
128 x 128), 20 launches in a HIP graph.  python tools/fixed_cost_probe.py [alternative library]"""
import ctypes, os, sys, torch
ROOT='/root/repo'; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests')); sys.path.insert(0, os.path.join(ROOT,'tools'))
from action_conditioned_gans_amd import _lib as L
from abi_call import Abi, _p
from conv16_probe import time_graph
lib=L.Library(sys.argv[1]) if len(sys.argv) > 1 else L.get(); abi=Abi(lib,'cuda:0',conv_dtype=L.ACG_BF16)
B,S,N=32,32,128
for k,cin in ((1,8),(1,64),(1,256),(3,64),(5,64)):
    d=abi.desc(B,S,S,cin,k,k,N,1,'SAME')
    x=torch.randn(B,S,S,cin,device='cuda').bfloat16(); w=torch.randn(k,k,cin,N,device='cuda')*0.05
    rm,tr=abi.prep_weights(w); y=torch.zeros(B,d.out_h,d.out_w,N,dtype=torch.bfloat16,device='cuda')
    ws,n=abi.ws(lib.conv2d_workspace_bytes(ctypes.byref(d),L.CONV_FWD,L.ACG_BF16))
    sp=lib.conv2d_splits(ctypes.byref(d),L.CONV_FWD,L.ACG_BF16)
    fn=lambda: lib.conv2d_fwd(_p(x),_p(tr),_p(y),ctypes.byref(d),L.ACG_BF16,_p(ws),n,abi.stream())
    us=time_graph(fn); fl=2.0*B*S*S*k*k*cin*N
    print('fwd %dx%d Cin=%3d -> 128, M=32768: splits %d  %6.1f us  %6.1f TF/s  (K-steps %d)' % (k,k,cin,sp,us,fl/us/1e6,k*k*((cin+63)//64) if cin>=64 else k*k))
