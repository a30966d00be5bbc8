"""bench.py's pieces that need no GPU: the live counter passes (roofline.traffic) against a stand-in profiler."""
import argparse
import os
import stat
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

FAKE = r"""#!/bin/bash
# stands in for rocprofv3: writes the counter file the real one would, with known values
while [ $# -gt 0 ]; do case "$1" in --pmc) c=$2; shift 2;; -d) d=$2; shift 2;; --) shift; break;; *) shift;; esac; done
echo "$@" >> "$FAKE_LOG"
test -n "$FAKE_FAIL" && exit 3
mkdir -p "$d/host"
f="$d/host/1_counter_collection.csv"
echo '"Dispatch_Id","Kernel_Name","Counter_Name","Counter_Value"' > "$f"
v=100; test "$c" = WRITE_SIZE && v=40
for i in 1 2 3 4; do echo "$i,\"void trtd::k_shade<31u, true, 512, 6>(trtd::SceneDev, trtd::ShadeArgs)\",\"$c\",$v.0" >> "$f"; done
echo "5,\"void trtd::k_trace_closest<false, 1, false, 0, true, 0>(trtd::SceneDev, trtd::RaySource)\",\"$c\",10.0" >> "$f"
echo "6,\"void trtd::k_trace_closest<false, 1, false, 0, false, 0>(trtd::SceneDev, trtd::RaySource)\",\"$c\",30.0" >> "$f"
echo "7,\"void trtd::k_trace_closest<true, 1, false, 0, true, 0>(trtd::SceneDev, trtd::RaySource)\",\"$c\",999.0" >> "$f"   # the counting build: left out
echo "8,\"__amd_rocclr_copyBuffer\",\"$c\",5.0" >> "$f"
"""


def _args(**kw):
    a = dict(scene="back", width=1920, height=1080, spp=256, builder="auto", tris=None, leaf=None, seed=None, fixed_nee=False)
    a.update(kw)
    return argparse.Namespace(**a)


def _install(tmp_path, monkeypatch):
    exe = tmp_path / "rocprofv3"
    exe.write_text(FAKE)
    exe.chmod(exe.stat().st_mode | stat.S_IXUSR)
    monkeypatch.setenv("PATH", f"{tmp_path}:{os.environ['PATH']}")
    monkeypatch.setenv("FAKE_LOG", str(tmp_path / "log.txt"))
    for k in [k for k in os.environ if k.startswith("ROCPROF") or k.startswith("ROCP_")]:
        monkeypatch.delenv(k)


def test_live_traffic_turns_the_counter_passes_into_bytes_per_launch(tmp_path, monkeypatch):
    """FETCH_SIZE / WRITE_SIZE are KiB summed per kernel; reads count twice (gfx950: a 128-B line is tallied as 64 B), the counting kernels of the
    untimed counting render are left out, the figure is per launch — exactly tools/pmc_summary.py's account of the committed files."""
    _install(tmp_path, monkeypatch)
    by, note = bench.live_traffic(_args(leaf=8, tris=5000))
    assert by == {"shade": (2 * 400 + 160) * 1024 // 4, "trace_closest": (2 * 40 + 40) * 1024 // 2}, (by, note)
    assert note.startswith("live:")
    log = open(tmp_path / "log.txt").read().splitlines()
    assert len(log) == 2  # one child per counter, each the program itself behind `--` (no shell, no env wrapper)
    for line in log:
        assert line.split()[0] == sys.executable and "bench.py" in line.split()[1]
        assert "--no-traffic" in line and "--steps 1" in line and "--leaf 8" in line and "--tris 5000" in line


def test_live_traffic_reports_why_it_could_not_run(tmp_path, monkeypatch):
    _install(tmp_path, monkeypatch)
    monkeypatch.setenv("FAKE_FAIL", "1")
    by, note = bench.live_traffic(_args())
    assert by == {} and "failed" in note
    monkeypatch.delenv("FAKE_FAIL")
    monkeypatch.setenv("ROCPROFILER_SOMETHING", "1")   # this process is itself being profiled: no nested passes
    by, note = bench.live_traffic(_args())
    assert by == {} and "not attempted" in note
