# Convenience targets (the driver uses __graft_entry__.py / pytest / bench.py directly).
.PHONY: build test test-gpu bench clean
build:
	python __graft_entry__.py
test: build
	python -m pytest tests -x -q -m "not gpu"
test-gpu: build
	python -m pytest tests -x -q -m gpu
bench: build
	python bench.py
clean:
	$(MAKE) -C motion-estimated-video-trimmer_amd/csrc clean
	$(MAKE) -C oracle clean
