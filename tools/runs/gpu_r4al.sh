R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4al; mkdir -p $O
cd $R
cat > /tmp/b1attn.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from mj_video_amd import ops
dev="cuda"; BF=torch.bfloat16
def run(name, n_seq, L, H, G, D, causal, mode, kernel):
    N=n_seq*L
    q=torch.randn(N,H*D,device=dev).to(BF); k=torch.randn(N,(H//G)*D,device=dev).to(BF); v=torch.randn(N,(H//G)*D,device=dev).to(BF); o=torch.empty(N,H*D,device=dev,dtype=BF)
    cu=torch.arange(0,(n_seq+1)*L,L,dtype=torch.int32,device=dev)
    for _ in range(5): ops.attention(q,k,v,o,cu,L,H,G,D,causal,D**-0.5,mode,kernel=kernel)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.attention(q,k,v,o,cu,L,H,G,D,causal,D**-0.5,mode,kernel=kernel)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:28s} kernel {kernel}: {e0.elapsed_time(e1)/50*1000:8.1f} us", flush=True)
for rep in range(2):
    for kern in (0, 6):
        run("vit b1: 8 x 1025, D=64", 8, 1025, 16, 1, 64, False, 2, kern)
        run("llm b1: 1 x 2186, D=128 causal", 1, 2186, 16, 2, 128, True, 2, kern)
PY
python /tmp/b1attn.py 2>/dev/null | tee $O/b1_attention.txt
