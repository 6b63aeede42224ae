"""Stand-in for `torch.distributed.run` in tests/test_bench_cli.py (TV_BENCH_LAUNCH_MODULE=_record_launch): records how bench.py's
self-launch called the launcher -- argv, whether torch was imported in THIS (child) interpreter's parent is reported by the parent on
stderr -- prints one JSON line the way rank 0 would, and exits with TV_FAKE_LAUNCH_RC (default 0)."""
import json
import os
import sys

if __name__ == "__main__":
    rc = int(os.environ.get("TV_FAKE_LAUNCH_RC", "0"))
    if os.environ.get("TV_FAKE_LAUNCH_SILENT", "0") != "1":
        print(json.dumps({"recorded_argv": sys.argv[1:], "self_launched": os.environ.get("TV_BENCH_SELF_LAUNCHED"),
                          "ipc_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), "backend": os.environ.get("TV_BENCH_BACKEND"), "torch_in_child": "torch" in sys.modules}), flush=True)
    if os.environ.get("TV_FAKE_LAUNCH_SLEEP"):
        import time
        time.sleep(float(os.environ["TV_FAKE_LAUNCH_SLEEP"]))
    sys.exit(rc)
