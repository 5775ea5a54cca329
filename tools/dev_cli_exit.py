"""Reproduces the round-2 driver failure: `hagrid_cli <obj> -r <rays> -n 3 -w 1 -k -nb 2` printed its report and never exited
while the parent Python process held a context on the same GPU.  Runs the command N times with a short timeout; on a timeout dumps
the kernel-side wait channel, the user stacks (rocgdb) and the HIP log tail of a re-run, then kills the process group.

usage: python tools/dev_cli_exit.py [N] [--no-parent-ctx] [--rocm-hip]
"""
import os
import signal
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _unused_dump_stacks(pid, out):
    for t in sorted(os.listdir(f"/proc/{pid}/task")):
        try:
            wchan = open(f"/proc/{pid}/task/{t}/wchan").read()
            st = [l for l in open(f"/proc/{pid}/task/{t}/status") if l.startswith(("Name", "State"))]
            out.write(f"  task {t}: wchan={wchan} {' '.join(s.strip() for s in st)}\n")
            try:
                out.write("    kstack: " + open(f"/proc/{pid}/task/{t}/stack").read().replace("\n", " | ") + "\n")
            except Exception as e:
                out.write(f"    kstack: {e}\n")
        except Exception as e:
            out.write(f"  task {t}: {e}\n")
    gdb = "/opt/rocm/bin/rocgdb"
    if os.path.exists(gdb):
        try:
            r = subprocess.run([gdb, "-p", str(pid), "-batch", "-ex", "set pagination off", "-ex", "thread apply all bt 30"],
                               capture_output=True, text=True, timeout=120)
            out.write(r.stdout[-12000:] + "\n" + r.stderr[-3000:] + "\n")
        except Exception as e:
            out.write(f"  rocgdb: {e}\n")
    out.flush()


def run_once(cmd, timeout, out, env=None):
    t0 = time.time()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True, env=env)
    try:
        so, se = p.communicate(timeout=timeout)
        return p.returncode, time.time() - t0, so, se
    except subprocess.TimeoutExpired:
        out.write(f"TIMEOUT after {timeout}s: {' '.join(cmd)}\n")
        import _subproc
        out.write(_subproc.thread_report(p.pid)); out.flush()
        os.killpg(p.pid, signal.SIGKILL)
        so, se = p.communicate()
        out.write("stdout tail: " + so[-600:] + "\nstderr tail: " + se[-3000:] + "\n")
        return None, time.time() - t0, so, se


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 30
    parent_ctx = "--no-parent-ctx" not in sys.argv
    import numpy as np
    from hagrid_amd import api, scene
    import test_cpp_api as T
    out = sys.stdout
    d = tempfile.mkdtemp()
    exe = T._build_cli(d, "/opt/rocm/lib" if "--rocm-hip" in sys.argv else None)
    tris = scene.make_soup(5000)
    v0 = tris[:, 0:3]; v1 = v0 - tris[:, 4:7]; v2 = v0 + tris[:, 8:11]
    obj = os.path.join(d, "soup.obj")
    with open(obj, "w") as f:
        for a, b, c in zip(v0, v1, v2):
            for p in (a, b, c):
                f.write("v %r %r %r\n" % (float(p[0]), float(p[1]), float(p[2])))
        for i in range(tris.shape[0]):
            f.write(f"f {3*i+1} {3*i+2} {3*i+3}\n")
    lo, hi = scene.tris_bbox(tris)
    rays = scene.make_rays_incoherent(lo, hi, 50000, 5)
    rfile = os.path.join(d, "soup.rays")
    np.ascontiguousarray(rays[:, [0, 1, 2, 4, 5, 6]]).tofile(rfile)
    mem = None
    if parent_ctx:
        mem = api.MemManager(keep=True)
        d_tris = mem.upload(tris)
        grid = api.build_all(mem, d_tris, tris.shape[0])
        d_rays = mem.upload(rays); d_hits = mem.alloc(16 * rays.shape[0])
        api.traverse_grid(grid, d_tris, d_rays, d_hits, rays.shape[0])
        mem.download(d_hits, api.HIT_DTYPE, rays.shape[0])
    cmds = [
        [exe, obj, "-r", rfile, "-n", "3", "-w", "1", "-k", "-nb", "2"],
        [exe, "soup:20000", "-z", "-sx", "256", "-sy", "128", "-o", os.path.join(d, "f.pgm")],
        [exe, obj, "-r", rfile, "-k", "--any-hit"],
        [exe, obj, "-r", rfile, "-k", "--gpus", "1", "-n", "2"],
    ]
    if "--only-first" in sys.argv:
        cmds = cmds[:1]
    hangs = 0
    worst = 0.0
    for i in range(n):
        for c in cmds:
            rc, dt, so, se = run_once(c, 40, out)
            worst = max(worst, dt)
            if rc is None:
                hangs += 1
                # once more with the runtime's log, to see the last API calls before the hang (if it hangs again)
                env = dict(os.environ, AMD_LOG_LEVEL="3")
                rc2, dt2, so2, se2 = run_once(c, 40, out, env)
                out.write(f"re-run with AMD_LOG_LEVEL=3: rc={rc2} {dt2:.1f}s\n" + se2[-4000:] + "\n")
            elif rc != 0:
                out.write(f"rc={rc}: {' '.join(c)}\n{so[-500:]}\n{se[-1500:]}\n")
        if i % 5 == 0:
            out.write(f"round {i}: hangs so far {hangs}, slowest call {worst:.2f}s\n"); out.flush()
    out.write(f"done: {n} rounds x {len(cmds)} commands, {hangs} hangs, slowest {worst:.2f}s\n")
    if mem is not None:
        mem.close()


if __name__ == "__main__":
    main()
