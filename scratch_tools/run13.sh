cd $GRAFT_REPO_ROOT
python - <<'PY'
import torch, time
n=1<<30
h=torch.empty(n,dtype=torch.uint8).pin_memory()
d=torch.empty(n,dtype=torch.uint8,device='cuda')
for _ in range(2): d.copy_(h,non_blocking=True); torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(5): d.copy_(h,non_blocking=True)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/5
print("pinned H2D GB/s", n/dt/1e9)
p=torch.empty(n,dtype=torch.uint8)
t0=time.perf_counter()
for _ in range(3): d.copy_(p); 
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/3
print("pageable H2D GB/s", n/dt/1e9)
PY
