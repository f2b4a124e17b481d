# Convenience targets; the driver's contract is __graft_entry__.build() / bench.py / pytest.
.PHONY: all build test test-gpu bench tools clean
all: build
build:
	python -c "import __graft_entry__ as g; g.build()"
test: build
	python -m pytest tests -x -q -m "not gpu"
test-gpu:
	python -m pytest tests -x -q -m gpu
bench:
	python bench.py
# developer probes (tools/README.md); binaries land in build/ (git-ignored, travels with gpurun)
TOOLS := symbench balbench kbench f64bench f64shapes bal_sim dp_mb valu_mb mfma_mb rsq64_probe pkbank_mb clock_probe
tools: build
	mkdir -p build
	for t in $(TOOLS); do hipcc -O3 -std=c++17 --offload-arch=gfx950 -I n-bodysimulation_amd/csrc -I include tools/$$t.hip -o build/$$t || exit 1; done
	hipcc -O2 -std=c++17 --offload-arch=gfx950 -I include tools/sync_probe.hip -o build/sync_probe -L n-bodysimulation_amd -lnbody_hip -Wl,-rpath,'$$ORIGIN/../n-bodysimulation_amd'
clean:
	$(MAKE) -C n-bodysimulation_amd/csrc clean
	$(MAKE) -C oracle clean
	rm -rf build
