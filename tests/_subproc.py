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


def _one_process(pid):
    lines = []
    try:
        tasks = sorted(os.listdir(f"/proc/{pid}/task"))
        cmd = open(f"/proc/{pid}/cmdline").read().replace("\0", " ")[:200]
    except OSError as e:
        return [f"/proc/{pid}: {e}"]
    lines.append(f" process {pid}: {cmd}")
    for t in tasks:
        try:
            wchan = open(f"/proc/{pid}/task/{t}/wchan").read()
            st = [l.strip() for l in open(f"/proc/{pid}/task/{t}/status") if l.startswith(("Name", "State"))]
            lines.append(f"  task {t}: wchan={wchan} {' '.join(st)}")
            try:
                lines.append("    kstack: " + open(f"/proc/{pid}/task/{t}/stack").read().replace("\n", " | "))
            except OSError:
                pass                                             # (not readable on the GPU box)
        except OSError as e:
            lines.append(f"  task {t}: {e}")
    return lines


def thread_report(pid):
    """Every process of the session `pid` leads (the front-end re-executes itself once per rank): command line, and per thread the kernel
    wait channel; user stacks of the leader through rocgdb where ptrace is allowed."""
    lines = []
    pids = [pid]
    for d in os.listdir("/proc"):
        if d.isdigit() and int(d) != pid:
            try:
                if os.getsid(int(d)) == pid:
                    pids.append(int(d))
            except OSError:
                pass
    for q in pids:
        lines += _one_process(q)
    gdb = "/opt/rocm/bin/rocgdb"
    if os.path.exists(gdb):
        for q in pids[:3]:
            try:
                r = subprocess.run([gdb, "-p", str(q), "-batch", "-ex", "set pagination off", "-ex", "thread apply all bt 30"],
                                   capture_output=True, text=True, timeout=120)
                if "Operation not permitted" in r.stderr:
                    lines.append("  rocgdb: ptrace not permitted here"); break
                lines.append(r.stdout[-8000:])
            except Exception as e:                               # diagnosis must never fail the caller
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
