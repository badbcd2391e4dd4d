# Top-level build for C users (the Python build in cuda_satabsearch_amd/build.py does the same):
#   make            libsatabsearch.so (HIP kernels + C ABI), libsathost.so (reader, statistics),
#                   cuda_satabsearch_amd/bin/satabsearch (command line)
#   make oracle     the CPU oracle used by the tests (oracle/), plus the reference build when
#                   /root/reference is mounted
#   make test       CPU test suite;  make test-gpu on a machine with an MI355X
HIPCC   ?= hipcc
CC      ?= gcc
PKG      = cuda_satabsearch_amd
CSRC     = $(PKG)/csrc
HOST     = $(CSRC)/host

all: $(PKG)/libsathost.so $(PKG)/libsatabsearch.so $(PKG)/bin/satabsearch

$(PKG)/libsathost.so: $(HOST)/sat_parse.c $(HOST)/sat_gumbel.c $(HOST)/sat_shard.c $(HOST)/sat_parse.h $(HOST)/sat_gumbel.h $(HOST)/sat_shard.h
	$(CC) -O2 -fPIC -shared -Wall -Wextra -I$(HOST) -o $@ $(HOST)/sat_parse.c $(HOST)/sat_gumbel.c $(HOST)/sat_shard.c -lm -lpthread

$(PKG)/sat_shard.o: $(HOST)/sat_shard.c $(HOST)/sat_shard.h
	$(CC) -O2 -fPIC -ffp-contract=off -Wall -Wextra -I$(HOST) -c -o $@ $(HOST)/sat_shard.c

$(PKG)/sat_gumbel.o: $(HOST)/sat_gumbel.c $(HOST)/sat_gumbel.h
	$(CC) -O2 -fPIC -ffp-contract=off -Wall -Wextra -I$(HOST) -c -o $@ $(HOST)/sat_gumbel.c

$(PKG)/libsatabsearch.so: $(CSRC)/sat_capi.hip $(CSRC)/sat_topk.hip $(CSRC)/sat_multi.hip $(PKG)/sat_gumbel.o $(PKG)/sat_shard.o $(CSRC)/sat_sa_kernel.hpp $(CSRC)/sat_ctx.hpp include/satabsearch.h
	$(HIPCC) --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared -Iinclude -I$(CSRC) -o $@ $(CSRC)/sat_capi.hip $(CSRC)/sat_topk.hip $(CSRC)/sat_multi.hip -Wl,$(PKG)/sat_gumbel.o -Wl,$(PKG)/sat_shard.o -lm -ldl

$(PKG)/bin/satabsearch: $(HOST)/sat_main.c $(HOST)/sat_host_search.c $(PKG)/libsatabsearch.so $(PKG)/libsathost.so
	mkdir -p $(PKG)/bin
	$(CC) -O3 -ffp-contract=off -Wall -Wextra -Iinclude -I$(HOST) -o $@ $(HOST)/sat_main.c $(HOST)/sat_host_search.c \
	    -L$(PKG) -lsatabsearch -lsathost -lm -Wl,-rpath,'$$ORIGIN/..'

oracle:
	$(MAKE) -C oracle all
	if [ -d /root/reference/nvcc_src_current ]; then $(MAKE) -C oracle ref; fi

test: all oracle
	python -m pytest tests -q -m "not gpu"

test-gpu: all oracle
	python -m pytest tests -q -m gpu

clean:
	rm -f $(PKG)/libsathost.so $(PKG)/libsatabsearch.so $(PKG)/bin/satabsearch
	$(MAKE) -C oracle clean

.PHONY: all oracle test test-gpu clean
