import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_trainer_gpu import Batches, c1_config, make

def run(ep, precision, save=False, ckpt=None):
    m, t = make(c1_config(ep=ep, save=save, precision=precision))
    t.fit(m, Batches(48, 1), Batches(24, 2), ckpt_path=ckpt)
    return m, t

def diff(a, b):
    return max(float((p.cpu() - q.cpu()).abs().max()) for p, q in zip(a.model.state_dict().values(), b.model.state_dict().values()))

for prec in ("32", "bf16-mixed"):
    os.environ["CKPT_DIR"] = tempfile.mkdtemp()
    a, _ = run(3, prec)
    b, _ = run(3, prec)
    print(prec, "two fresh 3-epoch runs differ by", diff(a, b))
    c, tc = run(2, prec, save=True)
    a2, _ = run(2, prec)
    print(prec, "two fresh 2-epoch runs differ by", diff(c, a2))
    d, td = run(3, prec, ckpt=tc.checkpointer.last_path)
    print(prec, "resumed vs straight:", diff(a, d), "steps", td.global_step, "opt step", td.optimizer._step,
          "dropout step", d.model.engine.step_counter, a.model.engine.step_counter)
