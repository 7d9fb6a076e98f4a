import ctypes as C, json, os, sys, time
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"rogue-gym_amd"))
import numpy as np, torch
from rogue_gym_python import _rogue_gym as inner
G=json.load(open(os.path.join(ROOT,"tests/golden/reference_goldens.json")))
n=65536
h=inner._Handle([json.dumps(dict(G["configs"]["mini"],seed=i)) for i in range(n)],1000,True)
L=h.L
dev=torch.device("cuda",0)
table=torch.tensor(list(b".hjklnbuy>s"),dtype=torch.uint8,device=dev)
keys=table[torch.randint(0,11,(64,n),device=dev)].contiguous()
torch.cuda.synchronize()
t0=time.time()
for t in range(300):
    L.rg_step(h.h, C.c_void_p(keys[t%64].data_ptr()),1)
    if t%50==49:
        torch.cuda.synchronize()
        print(t, "elapsed ms", (time.time()-t0)*1e3)
torch.cuda.synchronize()
