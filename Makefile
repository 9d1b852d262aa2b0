# The reference's entry points (its Makefile: install / compile / test / artifacts) for this build.  `make test` is the reference's `go test -v ./...` -- its three tests
# (zk_census_test.go) as the compiled client tests/host/reference_test_shape.cc over include/zkcensus_prover.hpp -- run from this directory against ./artifacts, with
# the same environment variables (CIRCUIT_NAME, ENVIRONMENT, NLEVELS, KEYSIZE, PADDING).  It needs an MI355X.  `make artifacts` fills ./artifacts with the build's own
# test key (the reference's proving_key.zkey is a blob it does not ship; with that file in place of this one nothing else changes).
NLEVELS ?= 160
CIRCUIT_NAME ?= zkCensus
ENVIRONMENT ?= dev
ART := artifacts/$(CIRCUIT_NAME)/$(ENVIRONMENT)/$(NLEVELS)
LIBDIR := zk-franchise-proof-circuit_amd

build:
	python -c "import __graft_entry__ as g; g.build()"

build/reference_test_shape: tests/host/reference_test_shape.cc include/zkcensus_prover.hpp include/zkcensus.h
	@mkdir -p build
	g++ -std=c++17 -O1 -Wall -Wextra $< -Iinclude -L$(LIBDIR) -lzkcensus -Wl,-rpath,$(abspath $(LIBDIR)) -o $@

artifacts: build
	@mkdir -p $(ART)
	python -c "import shutil; from zkcensus_amd import setup; r, z, v = setup.ensure_test_artifacts($(NLEVELS)); shutil.copyfile(z, '$(ART)/proving_key.zkey'); shutil.copyfile(v, '$(ART)/verification_key.json')"

test: build/reference_test_shape
ifeq (, $(wildcard ./artifacts/))
	$(error "run 'make artifacts' first")
else
	NLEVELS=$(NLEVELS) CIRCUIT_NAME=$(CIRCUIT_NAME) ENVIRONMENT=$(ENVIRONMENT) ./build/reference_test_shape
endif

check:
	python -m pytest tests -q -m "not gpu"

check-gpu:
	python -m pytest tests -q -m gpu

bench:
	python bench.py

.PHONY: build artifacts test check check-gpu bench
