"""Manual check (not collected by pytest): the row-sharded library test with 4 and 6 processes on the one GPU over the
peer-buffer transport.  Run on its own — this parent never touches the GPU, so 6 ranks stay within the box's limit of 6
GPU processes (inside pytest the session's own library instance would be the 7th)."""
import sys, tempfile, pathlib
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, "tests/multirank")
import test_multirank_gpu as t
for world in (4, 6):
    d = pathlib.Path(tempfile.mkdtemp())
    t.test_row_sharded_library_in_separate_processes(d, "ipc", world)
    print("world", world, "ok", flush=True)
