# Convenience targets; the driver's contract is __graft_entry__.build() / bench.py / pytest.
.PHONY: all build test test-gpu bench clean
all: build
build:
	python -c "import __graft_entry__ as g; g.build()"
test: build
	python -m pytest tests -x -q -m "not gpu"
test-gpu:
	python -m pytest tests -x -q -m gpu
bench:
	python bench.py
clean:
	$(MAKE) -C n-bodysimulation_amd/csrc clean
	$(MAKE) -C oracle clean
	rm -rf build
