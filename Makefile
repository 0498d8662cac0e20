# Builds everything in-tree (the .so files travel to the GPU box with the snapshot).
#   make            host library + HIP library + oracle (+ reference dump if available)
#   make hip        libpathed_hip.so only (hipcc, gfx950)
# __graft_entry__.build() runs `make all`.

CXX      ?= g++
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950

# -ffp-contract=off everywhere: the oracle and the kernels must round identically,
# fused multiply-adds appear only where the source says fmaf().
# x86-64-v3 (AVX2+FMA) rather than -march=native: the objects are built in one
# container and run on another host.
CPUFLAGS  = -std=c++17 -O2 -fPIC -ffp-contract=off -fwrapv -march=x86-64-v3 -Wall -Wextra -Iinclude
# (-instcombine-max-copied-from-constant-users: the kernels take RenderParams, 1.6 KB, by value; past 300 uses of it in one
#  kernel the compiler stops reading the fields from the kernel-argument segment and gives every lane a private copy in scratch)
HIPFLAGS  = -std=c++17 -O3 -fPIC -ffp-contract=off --offload-arch=$(ARCH) -Iinclude -Wall -Wno-unused-parameter \
            -mllvm -instcombine-max-copied-from-constant-users=4000

LIBDIR    = pathed_amd/lib
BINDIR    = pathed_amd/bin

HOST_SRC  = $(wildcard pathed_amd/host/*.cpp)
HOST_LIB_SRC = $(filter-out pathed_amd/host/main.cpp,$(HOST_SRC))
HOST_HDR  = $(wildcard pathed_amd/host/*.h) include/pathed_hip.h
HIP_SRC   = $(wildcard pathed_amd/csrc/*.hip)
HIP_HDR   = $(wildcard pathed_amd/csrc/*.h) include/pathed_hip.h

.PHONY: all host hip oracle ref assets experiments clean

all: hip host oracle ref assets

hip: $(LIBDIR)/libpathed_hip.so
host: $(LIBDIR)/libpathed_host.so $(BINDIR)/pathed
oracle: oracle/liboracle.so

$(LIBDIR)/libpathed_hip.so: $(HIP_SRC) $(HIP_HDR) Makefile
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(HIP_SRC)

# The measured-and-rejected kernel organisations (kernels_experiments.h: staged and split shade stages, compressed node
# formats, the matrix-pipe phase 1, the clocked VALU probe) are NOT part of `make all`.  Load the result with
# PATHED_HIP_LIB=pathed_amd/lib/libpathed_hip_experiments.so (pathed_amd/_capi.py); tests: pytest -m experiments.
experiments: $(LIBDIR)/libpathed_hip_experiments.so
$(LIBDIR)/libpathed_hip_experiments.so: $(HIP_SRC) $(HIP_HDR) Makefile
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -DPATHED_EXPERIMENTS=1 -shared -o $@ $(HIP_SRC)

$(LIBDIR)/libpathed_host.so: $(HOST_LIB_SRC) $(HOST_HDR) $(LIBDIR)/libpathed_hip.so
	@mkdir -p $(LIBDIR)
	$(CXX) $(CPUFLAGS) -shared -o $@ $(HOST_LIB_SRC) -L$(LIBDIR) -lpathed_hip -lz -lpthread -Wl,-rpath,'$$ORIGIN'

$(BINDIR)/pathed: pathed_amd/host/main.cpp $(LIBDIR)/libpathed_host.so
	@mkdir -p $(BINDIR)
	$(CXX) $(CPUFLAGS) -o $@ pathed_amd/host/main.cpp -L$(LIBDIR) -lpathed_host -lpathed_hip -lpthread -Wl,-rpath,'$$ORIGIN/../lib'

oracle/liboracle.so: oracle/oracle.cpp oracle/oracle.h include/pathed_hip.h
	$(CXX) $(CPUFLAGS) -fopenmp -shared -o $@ oracle/oracle.cpp

# The reference's own translation units (only those that compile without Embree),
# linked with oracle/ref_driver.cpp into a dump tool.  Needs /root/reference, which
# exists in the build container only; the output stays out of git (oracle/_ref/).
ref:
	@if [ -d /root/reference/src ]; then $(MAKE) -C oracle -f Makefile.ref; else echo "reference sources absent: skipping oracle/_ref"; fi

# synthetic stand-ins for the assets the reference's scene files point at but does not ship
assets: host
	python3 tools/make_assets.py

clean:
	rm -rf $(LIBDIR) $(BINDIR) oracle/liboracle.so oracle/_ref
