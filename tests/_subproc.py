"""Runs a front-end binary with a short time limit; a process that does not exit is diagnosed (kernel wait channel and
stack of every thread, user stacks through rocgdb when present) and its whole process group killed, so a front-end fault
costs one test a minute instead of masking the kernel-parity tests behind it."""
import os
import signal
import subprocess
import time


class Result:
    def __init__(self, returncode, stdout, stderr, seconds, diagnosis=""):
        self.returncode, self.stdout, self.stderr, self.seconds, self.diagnosis = returncode, stdout, stderr, seconds, diagnosis

    @property
    def timed_out(self):
        return self.returncode is None


def thread_report(pid):
    lines = []
    try:
        tasks = sorted(os.listdir(f"/proc/{pid}/task"))
    except OSError as e:
        return f"/proc/{pid}: {e}\n"
    for t in tasks:
        try:
            wchan = open(f"/proc/{pid}/task/{t}/wchan").read()
            st = [l.strip() for l in open(f"/proc/{pid}/task/{t}/status") if l.startswith(("Name", "State"))]
            lines.append(f"  task {t}: wchan={wchan} {' '.join(st)}")
            try:
                lines.append("    kstack: " + open(f"/proc/{pid}/task/{t}/stack").read().replace("\n", " | "))
            except OSError as e:
                lines.append(f"    kstack: {e}")
        except OSError as e:
            lines.append(f"  task {t}: {e}")
    gdb = "/opt/rocm/bin/rocgdb"
    if os.path.exists(gdb):
        try:
            r = subprocess.run([gdb, "-p", str(pid), "-batch", "-ex", "set pagination off", "-ex", "thread apply all bt 30"],
                               capture_output=True, text=True, timeout=120)
            lines.append(r.stdout[-12000:])
            lines.append(r.stderr[-2000:])
        except Exception as e:                                   # diagnosis must never fail the caller
            lines.append(f"  rocgdb: {e}")
    return "\n".join(lines) + "\n"


def run(cmd, timeout=60, env=None, cwd=None):
    t0 = time.time()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True, env=env, cwd=cwd)
    try:
        so, se = p.communicate(timeout=timeout)
        return Result(p.returncode, so, se, time.time() - t0)
    except subprocess.TimeoutExpired:
        diag = f"TIMEOUT after {timeout}s: {' '.join(map(str, cmd))}\n" + thread_report(p.pid)
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass
        so, se = p.communicate()
        diag += "stdout tail: " + so[-1500:] + "\nstderr tail: " + se[-3000:] + "\n"
        return Result(None, so, se, time.time() - t0, diag)


def check(cmd, timeout=60, env=None):
    """run() that fails the calling test with the diagnosis on a timeout and with both streams on a non-zero exit."""
    r = run(cmd, timeout, env)
    assert not r.timed_out, r.diagnosis
    assert r.returncode == 0, f"exit code {r.returncode}: {' '.join(map(str, cmd))}\n{r.stdout[-2000:]}\n{r.stderr[-3000:]}"
    if "teardown did not finish" in r.stderr:                    # hagrid_cli's own watchdog ended a teardown that hung: results complete, but say so
        import warnings
        warnings.warn(f"{os.path.basename(str(cmd[0]))}: teardown watchdog fired after {r.seconds:.0f} s: {r.stderr[-300:]}")
    return r
