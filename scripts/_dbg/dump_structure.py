import os, sys
cfg = sys.argv[1]
os.environ["MAVBA_CHOL_DUMP"] = f"gpurun_out/struct_{cfg}.txt"
sys.path.insert(0, ".")
import mavmap_amd
from mavmap_amd import synth
p = synth.make_config(cfg)
with mavmap_amd.Session(p, {}) as s:
    print(cfg, s.info()["matrix_dim"], s.info()["chol_model_forward_us"])
