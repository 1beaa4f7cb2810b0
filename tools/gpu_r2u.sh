R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2u; mkdir -p $O
cd $R
python - > $O/eq.log 2>&1 <<'PY'
import sys; sys.path.insert(0,'.')
import torch, mj_video_amd
from mj_video_amd import ops
dev='cuda'; BF=torch.bfloat16
for (n_seq,L,H,G,D,causal,mode) in [(64,1025,16,1,64,False,0),(8,2186,16,2,128,True,1),(3,700,16,2,128,True,1)]:
    N=n_seq*L
    q=torch.randn(N,H*D,device=dev).to(BF); k=torch.randn(N,(H//G)*D,device=dev).to(BF); v=torch.randn(N,(H//G)*D,device=dev).to(BF)
    cu=torch.arange(0,(n_seq+1)*L,L,dtype=torch.int32,device=dev)
    outs=[]
    for var in (0,4,4,4):
        ops.attention_set_variant(var)
        o=torch.zeros(N,H*D,device=dev,dtype=BF)
        ops.attention(q,k,v,o,cu,L,H,G,D,causal,D**-0.5,mode)
        outs.append(o.clone())
    ops.attention_set_variant(0)
    print(D, L, [bool(torch.equal(outs[0],x)) for x in outs[1:]])
PY
cat $O/eq.log | tail -4
timeout 600 python tools/attn_bench.py 20 4 0,4 > $O/ab.log 2>&1; grep variant $O/ab.log
