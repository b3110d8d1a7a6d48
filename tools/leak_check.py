import torch, gc
from ikflow_amd.ikflow_solver import IKFlowSolver
from ikflow_amd.model import hparams_for, layout_from, random_state_dict
from ikflow_amd.robots import get_robot
name="panda__full__lp191_5.25m"
robot=get_robot("panda"); hp=hparams_for(name); lay=layout_from(hp,robot); sd=random_state_dict(lay,robot,0)
dev=torch.device("cuda",0)
poses=torch.randn(300,7,device=dev); poses[:,3:]/=poses[:,3:].norm(dim=1,keepdim=True)
def once(prec):
    s=IKFlowSolver(hp,robot); s.load_state_dict_tensors(sd); s.set_precision(prec)
    s.generate_ik_solutions(poses); s.generate_exact_ik_solutions(poses[:50])
    del s; gc.collect(); torch.cuda.synchronize()
once("f32")
free0=torch.cuda.mem_get_info()[0]
for i in range(12): once("f16x3" if i%2 else "f32")
free1=torch.cuda.mem_get_info()[0]
print("free before %.1f MB after %.1f MB delta %.1f MB" % (free0/1e6, free1/1e6, (free0-free1)/1e6))
