"""Which fork / join shapes does hipStreamEndCapture of this ROCm survive?  (r05: the ScoreMapModule side-stream experiment crashed the host
inside torch.cuda.graphs.capture_end; the shipped two-stream capture never has.)  Each case runs in a child process -- a segfault must not
take the others down -- and captures a few trivial kernels on streams forked from the capture's origin stream:
  two_forks        : s1, s2 fork from the origin, join the origin                                (the shipped Stepper's shape)
  cross_one_way    : + s3 forks from the origin, waits on an event of s1, joins the origin
  cross_join_back  : + s1 then waits on s3 (s3 joins INTO s1, s1 joins the origin)               (the side-stream experiment's shape)
  nested           : s3 enters the capture only by waiting on s1 (a fork of a fork), joins s1
  nested_join_origin: s3 enters by waiting on s1, joins the origin directly
  join_back_only   : s3 forks from the origin (no wait on s1); s1 waits on s3; the origin joins s1, s2 AND s3
  cross_join_back_origin : cross_join_back + the origin also joins s3 directly
  continue_on_side : s3 enters by waiting on s1, s1 works on, s3 waits on s1 AGAIN and carries on; the origin joins s1, s2, s3
                     (the shape the side-stream experiment was rebuilt on)
    python scripts/proto/capture_fork_probe.py            # runs every case, prints one line each
    python scripts/proto/capture_fork_probe.py CASE       # one case in this process"""
import subprocess
import sys

CASES = ["two_forks", "cross_one_way", "cross_join_back", "nested", "nested_join_origin", "join_back_only", "cross_join_back_origin", "continue_on_side"]


def run(case):
    import torch
    dev = torch.device("cuda", 0)
    a, b, c = (torch.zeros(1 << 20, device=dev) for _ in range(3))
    origin, s1, s2, s3 = (torch.cuda.Stream() for _ in range(4))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(origin):
        with torch.cuda.graph(g, stream=origin, capture_error_mode="thread_local"):
            s1.wait_stream(origin)
            s2.wait_stream(origin)
            if case in ("cross_one_way", "cross_join_back", "join_back_only", "cross_join_back_origin"):
                s3.wait_stream(origin)
            with torch.cuda.stream(s1):
                a.add_(1.0)
            with torch.cuda.stream(s2):
                b.add_(1.0)
            if case != "two_forks":
                if case != "join_back_only":
                    s3.wait_stream(s1)
                with torch.cuda.stream(s3):
                    c.add_(a if case != "join_back_only" else 1.0)
                with torch.cuda.stream(s1):
                    a.mul_(2.0)
                if case in ("cross_join_back", "nested", "join_back_only", "cross_join_back_origin"):
                    s1.wait_stream(s3)
                if case == "continue_on_side":
                    s3.wait_stream(s1)
                    with torch.cuda.stream(s3):
                        c.add_(a)
                if case not in ("cross_join_back", "nested"):
                    origin.wait_stream(s3)
            origin.wait_stream(s1)
            origin.wait_stream(s2)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("%s: captured and replayed; a = %.0f, b = %.0f, c = %.0f" % (case, float(a[0]), float(b[0]), float(c[0])), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for case in CASES:
            r = subprocess.run([sys.executable, "-X", "faulthandler", __file__, case], capture_output=True, text=True, timeout=300)
            where = [ln.strip() for ln in r.stderr.splitlines() if "File" in ln][:1]
            print("%-20s rc=%4d  %s %s" % (case, r.returncode, r.stdout.strip(), ("| crashed in: " + where[0]) if r.returncode and where else ""), flush=True)
