"""Host time of building a BatchPlan + its ttv_batch for ragged 4-7 clip batches (what the config-3 driver pays per step when the
plan cache misses).  DEVICE=cpu|cuda:0."""
import cProfile, os, pstats, random, time, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from titok_video_amd.plan import BatchPlan
random.seed(0)
shapes=[(16,128,128),(8,64,96),(16,64,64),(4,128,96),(12,96,128),(16,96,96)]
def make():
    n=random.randint(4,7)
    g=[random.choice(shapes) for _ in range(n)]
    c=[random.choice([32,64,128]) for _ in range(n)]
    return g,c
batches=[make() for _ in range(200)]
dev=torch.device(os.environ.get('DEVICE', 'cuda:0'))
def run():
    for g,c in batches:
        p=BatchPlan(g,c,(4,8,8),dev)
        p.batch_for(4,2)
t0=time.perf_counter(); run(); t1=time.perf_counter()
print('per plan+batch_for: %.3f ms'%((t1-t0)/len(batches)*1e3))
pr=cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
