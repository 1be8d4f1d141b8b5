import sys, time, os, torch
sys.path.insert(0,'/root/repo')
import co_occ_amd as pkg, co_occ_amd.synth as synth
from oracle import ref_cpu
torch.set_num_threads(int(sys.argv[1]))
c = synth.CONFIGS["r50"]
cfg = synth.model_cfg()
model = pkg.build_detector(cfg)
sd = synth.random_state_dict(model.state_dict(), seed=0)
img, pts = synth.voxel_inputs(c["grid"], C=128, seed=1234)
sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
t=time.perf_counter()
with torch.no_grad():
    o = ref_cpu.bifuser_fuse(sub("occ_fuser."), img, pts, 2); t1=time.perf_counter(); print("fuse", t1-t, flush=True)
    vf = ref_cpu.con_enc(sub("occ_fuser."), o["all_feats"]); t2=time.perf_counter(); print("con_enc", t2-t1, flush=True)
    mid = ref_cpu.resnet3d_forward(sub("semantic_encoder."), vf); t3=time.perf_counter(); print("resnet", t3-t2, flush=True)
    sem = ref_cpu.fpn3d_forward(sub("semantic_neck."), mid); t4=time.perf_counter(); print("fpn", t4-t3, flush=True)
