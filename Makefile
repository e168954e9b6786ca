# Convenience targets (the driver uses __graft_entry__.build(), bench.py and pytest directly).
PY ?= python

lib:
	$(PY) mellow_amd/csrc/build.py

test-cpu: lib
	$(PY) -m pytest tests -x -q -m "not gpu"

test-gpu: lib
	$(PY) -m pytest tests -x -q -m gpu

bench: lib
	$(PY) bench.py

microbench:
	for f in mfma_peak gemm_ablate grid_sync grid_sync2 stage_emul fused_og_emul; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/microbench/$$f.bin tools/microbench/$$f.hip; done

.PHONY: lib test-cpu test-gpu bench microbench
