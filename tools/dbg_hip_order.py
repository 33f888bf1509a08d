import sys, tempfile
sys.path.insert(0, '.'); sys.path.insert(0, 'globalsfmpy_amd')
mode = sys.argv[1]
if mode == "C":
    import torch; print("torch first", torch.cuda.is_available())
import GlobalSfMpy as sfm
from globalsfmpy_amd import dataset_1dsfm as ds
if mode == "B":
    import torch; print("torch second", torch.cuda.is_available())
tmp = tempfile.mkdtemp(); ds.write_synthetic_dataset(tmp)
try:
    print(mode, sfm.CalcCovariance(tmp))
except Exception as e:
    print(mode, "FAILED", e)

print(sorted(set(l.split()[-1] for l in open("/proc/self/maps") if "amdhip64" in l)))
