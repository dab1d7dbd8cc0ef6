# Builds every native artefact into build/ (git-ignored, but shipped to the GPU box by gpurun).
#   build/libsquid_hip.so   product: HIP kernels + host pipeline behind the C ABI of include/squid_hip.h
#   build/squid             product: drop-in command line
#   build/squid_junction    product: counterpart of utils/JunctionSequence.cpp (junction sequences of the calls of a _sv.txt)
#   build/squid_annotate    product: counterpart of utils/AnnotateSQUIDOutput.py (GTF join over _sv.txt)
#   build/gen_synth_bam     synthetic BAM generator (inputs for tests and bench)
#   build/squid_oracle      CPU oracle (test infrastructure; never linked into the product)
HIPCC ?= hipcc
CXX ?= g++
ARCH ?= gfx950
B := build
CSRC := squid_amd/csrc
LIBSRC := $(CSRC)/sq_kernels.hip $(CSRC)/sq_bam.cpp $(CSRC)/sq_chimeric.cpp $(CSRC)/sq_segment.cpp $(CSRC)/sq_graph.cpp $(CSRC)/sq_order.cpp $(CSRC)/sq_post.cpp $(CSRC)/sq_bwa.cpp $(CSRC)/sq_junction.cpp $(CSRC)/sq_exchange.cpp $(CSRC)/sq_capi.cpp

all: $(B)/libsquid_hip.so $(B)/squid $(B)/squid_junction $(B)/squid_annotate $(B)/gen_synth_bam $(B)/squid_oracle $(B)/oracle_singlebamrec ref

# oracle/_ref: the part of the real reference that builds without third-party libraries (flag parser), only when
# the reference tree is present (authoring container); the GPU box uses the prebuilt binary
ref:
	$(MAKE) -C oracle ref

$(B)/libsquid_hip.so: $(LIBSRC) $(CSRC)/sq_internal.h $(CSRC)/sq_parsort.h $(CSRC)/sq_graph_kernels.inc $(CSRC)/sq_pass_kernels.inc $(CSRC)/sq_inflate_spec.inc $(CSRC)/sq_resolve.inc $(CSRC)/sq_wave.h include/squid_hip.h
	mkdir -p $(B)
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wall -Wno-unused-function -o $@ $(LIBSRC) -lz -lpthread -ldl

$(B)/squid: $(CSRC)/squid_main.cpp $(B)/libsquid_hip.so include/squid_hip.h
	$(HIPCC) -O2 -std=c++17 -o $@ $(CSRC)/squid_main.cpp -L$(B) -lsquid_hip -Wl,-rpath,'$$ORIGIN'

$(B)/squid_junction: $(CSRC)/squid_junction.cpp $(B)/libsquid_hip.so include/squid_hip.h
	$(HIPCC) -O2 -std=c++17 -o $@ $(CSRC)/squid_junction.cpp -L$(B) -lsquid_hip -Wl,-rpath,'$$ORIGIN'

$(B)/squid_annotate: $(CSRC)/squid_annotate.cpp
	mkdir -p $(B)
	$(CXX) -O2 -std=c++17 -Wall -o $@ $<

$(B)/gen_synth_bam: squid_amd/synth/gen_synth_bam.cpp
	mkdir -p $(B)
	$(CXX) -O2 -std=c++17 -o $@ $< -lz -lpthread -ldl

$(B)/squid_oracle: oracle/squid_oracle.cpp oracle/o_bam.h oracle/o_readrec.h oracle/o_graph.h oracle/o_order.h oracle/o_post.h oracle/o_bwa.h oracle/o_junction.h
	$(MAKE) -C oracle OUT=../$(B)

$(B)/oracle_singlebamrec: oracle/singlebamrec_driver.cpp oracle/o_readrec.h oracle/o_bam.h
	$(MAKE) -C oracle OUT=../$(B) ../$(B)/oracle_singlebamrec

clean:
	rm -rf $(B)
.PHONY: all clean ref
