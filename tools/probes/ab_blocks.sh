for W in 1000 4000 1000 4000; do C8P_MAX_WAIT_US=$W python - <<'PY'
import os, sys, json
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import importlib.util
spec = importlib.util.spec_from_file_location("bb", "tools/bench_blocks.py"); bb = importlib.util.module_from_spec(spec); spec.loader.exec_module(bb)
r = bb.run(2048, batches=(256,), quiet=True)
print("wait_us", os.environ["C8P_MAX_WAIT_US"], {k: v for k, v in r.items() if k.endswith("per_s")}, json.dumps(r["chain_host_phase_us_per_packet"]))
PY
done
