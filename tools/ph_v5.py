import sys, os
sys.path.insert(0, os.getcwd())
import runpy
from lifelong_nnunet_amd import native as nat
nat.lib().lnn_debug_force_conv_kernel(5)
sys.argv = ["kbench.py", "--layers", "dec3.0,dec2.0,enc3.1", "--which", "fwd", "--iters", "5", "--phases"]
runpy.run_path("tools/kbench.py", run_name="__main__")
