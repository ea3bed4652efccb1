"""Child programs of the GPU tests are started by a small helper process instead of the test process itself.

The helper is started at the beginning of the session, before anything in the test process has touched the GPU,
and never touches it.  The test process initialises the HIP runtime early on; starting programs from it means
fork + exec in a process with a live GPU runtime, and one full run of the suite died of a segmentation fault
inside subprocess.Popen at exactly that point.  `run` has the shape of subprocess.run(..., capture_output=True,
text=True) for the few arguments the tests use.
"""
import atexit
import json
import os
import subprocess
import sys

_server = None


def start():
    global _server
    if _server is None or _server.poll() is not None:
        _server = subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__), "--serve"],
                                   stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
        atexit.register(stop)
    return _server


def stop():
    global _server
    if _server is not None and _server.poll() is None:
        try:
            _server.stdin.close()
            _server.wait(timeout=10)
        except Exception:
            _server.kill()
    _server = None


class Result(object):
    def __init__(self, d):
        self.returncode, self.stdout, self.stderr = d["returncode"], d["stdout"], d["stderr"]


def run(args, env=None, cwd=None, timeout=None):
    srv = start()
    srv.stdin.write(json.dumps({"args": [str(a) for a in args], "env": env, "cwd": cwd, "timeout": timeout}) + "\n")
    srv.stdin.flush()
    line = srv.stdout.readline()
    if not line:
        raise RuntimeError("the spawn helper has gone away")
    d = json.loads(line)
    if d.get("timeout"):
        raise subprocess.TimeoutExpired(args, timeout, output=d["stdout"], stderr=d["stderr"])
    return Result(d)


def _serve():
    for line in sys.stdin:
        req = json.loads(line)
        try:
            r = subprocess.run(req["args"], env=req["env"], cwd=req["cwd"], timeout=req["timeout"],
                               capture_output=True, text=True, errors="replace")
            out = {"returncode": r.returncode, "stdout": r.stdout, "stderr": r.stderr}
        except subprocess.TimeoutExpired as e:
            text = lambda b: b.decode(errors="replace") if isinstance(b, bytes) else (b or "")
            out = {"timeout": True, "returncode": None, "stdout": text(e.stdout), "stderr": text(e.stderr)}
        except OSError as e:
            out = {"returncode": 127, "stdout": "", "stderr": repr(e)}
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()


if __name__ == "__main__" and "--serve" in sys.argv:
    _serve()
