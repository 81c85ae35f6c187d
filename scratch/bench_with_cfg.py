"""bench.py under a dfw_config override: DFW_CFG="conv_patch=2" python scratch/bench_with_cfg.py [bench.py args]
(bench.py itself reads no tuning switches; this wrapper applies scratch/_cfg.py first and then runs it in-process)."""
import sys, os, runpy
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _cfg
print("[cfg]", _cfg.apply_env_config(), file=sys.stderr)
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
