# convenience targets; __graft_entry__.build() does the same from Python
HIPCC ?= hipcc
LIB    = quadsim_amd/csrc/libquadsim_hip.so
SRC    = quadsim_amd/csrc/quadsim_hip.hip
HDR    = $(wildcard quadsim_amd/csrc/*.hpp) include/quadsim.h

all: lib oracle

lib: $(LIB)
$(LIB): $(SRC) $(HDR)
	$(HIPCC) -std=c++20 -O3 -fno-slp-vectorize -ffp-contract=on --offload-arch=gfx950 -fPIC -shared -Wno-unused-result $(SRC) -lhsa-runtime64 -o $@

oracle:
	$(MAKE) -C oracle

test-cpu: all
	python -m pytest tests -q -m "not gpu"

test-gpu: all
	python -m pytest tests -q -m gpu

bench: all
	python bench.py

clean:
	rm -f $(LIB) oracle/libqso.so oracle/*.o

.PHONY: all lib oracle test-cpu test-gpu bench clean
