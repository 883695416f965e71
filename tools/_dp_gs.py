import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch, bench
print(json.dumps({k: v for k, v in bench.dp_gs_leg(0, 1, torch.device('cuda', 0)).items() if k in ('ms_per_step', 'hip_graph')}))
