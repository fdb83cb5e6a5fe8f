# Convenience targets; the real entry points are python -m tinman_sandbox_amd.build,
# pytest and bench.py (see README.md).
.PHONY: build test test-gpu bench clean
build:
	python -m tinman_sandbox_amd.build
	$(MAKE) -C oracle all
test: build
	python -m pytest tests -q -m "not gpu"
test-gpu: build
	python -m pytest tests -q -m gpu
bench: build
	python bench.py
clean:
	rm -f tinman_sandbox_amd/csrc/*.so tinman_sandbox_amd/host/*.so tinman_sandbox_amd/host/caar_driver \
	      tinman_sandbox_amd/host/caar_driver_np4_nlev128 tinman_sandbox_amd/host/caar_driver_np8_nlev72
	$(MAKE) -C oracle clean
