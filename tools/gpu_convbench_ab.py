"""A/B harness for kernel experiments (interleaved rounds in one process family on one box: boxes differ by +-4 %).
The variants are selected by the environment variable RTM3D_AB, which an experimental build of a launcher reads
with getenv(); the committed kernels read no environment variables."""
import sys, os, subprocess
here = os.path.dirname(os.path.abspath(__file__))
code = "import sys; sys.path.insert(0, %r); from gpu_convbench import one; r0 = one(32, 96, 320, 256, 1024, 3, 6, 0, reps=2); one(32, 96, 320, 256, 1024, 3, 6, 2, reps=10, check=r0)" % here
for rnd in range(3):
    for sched in sys.argv[1:] or ('0', '1'):
        env = dict(os.environ, RTM3D_AB=sched)
        out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True).stdout.strip().splitlines()
        print('sched', sched, out[-2][-60:], out[-1][-40:])
