R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2u; mkdir -p $O
cd $R
timeout 300 python - > $O/eq.log 2>&1 <<'PY'
import sys; sys.path.insert(0,'.')
import torch, mj_video_amd
from mj_video_amd import ops
dev='cuda'; BF=torch.bfloat16
torch.manual_seed(0)
for (n_seq,L,H,G,D,causal,mode) in [(64,1025,16,1,64,False,0),(8,2186,16,2,128,True,1),(5,1,16,1,64,False,0),(4,33,16,1,64,False,0),(6,130,16,1,64,False,0),(3,257,16,1,64,True,0),(2,129,16,2,128,True,1),(2,5000,16,2,128,True,1)]:
    N=n_seq*L
    q=torch.randn(N,H*D,device=dev).to(BF); k=torch.randn(N,(H//G)*D,device=dev).to(BF); v=torch.randn(N,(H//G)*D,device=dev).to(BF)
    cu=torch.arange(0,(n_seq+1)*L,L,dtype=torch.int32,device=dev)
    outs=[]
    for var in (4,0,0,0):
        ops.attention_set_variant(var)
        o=torch.zeros(N,H*D,device=dev,dtype=BF)
        ops.attention(q,k,v,o,cu,L,H,G,D,causal,D**-0.5,mode)
        torch.cuda.synchronize()
        outs.append(o.clone())
    ops.attention_set_variant(0)
    print(D, L, causal, [bool(torch.equal(outs[0],x)) for x in outs[1:]])
PY
cat $O/eq.log | tail -9
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attention" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest.log
timeout 600 python tools/attn_bench.py 20 4 0,4 > $O/ab.log 2>&1; grep variant $O/ab.log | sort | awk '{k=$1" "$2" "$3; if(min[k]==""||$4<min[k])min[k]=$4} END{for(k in min) print k, "min", min[k]}' | sort -k3,3 -k2,2n
