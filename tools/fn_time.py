import sys, os
sys.path[:0]=[os.environ.get("GRAFT_REPO_ROOT","/root/repo"), os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"monkey-net_amd")]
import torch
from mnk import ops
from modules.util import ResBlock3D
dev=torch.device("cuda:0")
torch.manual_seed(0)
b0=ResBlock3D(45,kernel_size=(1,3,3),padding=(0,1,1)).to(dev).train()
b1=ResBlock3D(45,kernel_size=(1,3,3),padding=(0,1,1)).to(dev).train()
x=ops.to_act(torch.rand(32,45,1,64,64,device=dev))
def run():
    with torch.no_grad():
        out,c,s=b0.forward_act(x,45,want_stats=True,next_norm=b1.norm1)
        out,c=b1.forward_act(out,c,x_sums=s)
    return out
for _ in range(3): run()
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
g=torch.cuda.CUDAGraph()
s_=torch.cuda.Stream(); s_.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s_):
    run()
torch.cuda.current_stream().wait_stream(s_)
with torch.cuda.graph(g):
    run()
for _ in range(3): g.replay()
torch.cuda.synchronize()
e0.record()
for _ in range(50): g.replay()
e1.record(); torch.cuda.synchronize()
print("two residual blocks forward: %.1f us per pass (fused launches %s, error %d)" % (e0.elapsed_time(e1)/50*1e3, ops.FUSED_NORM_COUNT, ops.fused_norm_error()))
