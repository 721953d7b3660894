"""The reference's "deep" evaluation (scrambling_depths=range(0): 100-999 random moves per game), 500 games, CLI defaults."""
import os, sys, time, json
import numpy as np, torch
ROOT="/root/repo"
sys.path[:0]=[ROOT, os.path.join(ROOT,"rl-rubiks_amd")]
from librubiks.model import Model
from librubiks.solving.agents import MCTS, AStar
from librubiks.solving.evaluation import Evaluator
from librubiks.utils import set_seeds
model=Model.load(os.path.join(ROOT,"weights","fc_small_r1")).eval()
out={}
for name, agent in (("mcts", MCTS(model, c=0.6, search_graph=True)), ("astar", AStar(model, lambda_=0.2, expansions=100))):
    set_seeds()
    ev=Evaluator(500, range(0), None, 175_000, slots=1024 if name=="mcts" else None)
    torch.cuda.synchronize(); t=time.perf_counter()
    res, states, times = ev.eval(agent)
    torch.cuda.synchronize(); dt=time.perf_counter()-t
    s=ev.log_this_depth(res[0], states[0], times[0], 0)
    out[name]={"seconds":dt, **{k:s[k] for k in ("share_completed","mean_turns","states_per_game")}}
    print(name, json.dumps(out[name]), flush=True)
    del agent; torch.cuda.empty_cache()
