import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
import oracle_lib
from megaverse_amd.extension import MegaverseGym
scn = "ObstaclesEasy" if "o" in mode else "TowerBuilding"
if "t" in mode:
    import torch; print("torch avail", torch.cuda.is_available())
if "O" in mode:
    og = oracle_lib.OracleGym(scn, 32, 32, 4, 1, 1, False, None); og.seed(1); og.reset()
try:
    hg = MegaverseGym(scn, 32, 32, 4, 1, 1, False, {}); hg.seed(1); hg.reset(); print(mode, "OK")
except Exception as e:
    print(mode, "FAIL", e)
