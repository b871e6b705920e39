# Builds the C-ABI kernel library for MI355X (gfx950) and the oracle's native pieces.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := tmdiff_amd/csrc
SRCS  := $(CSRC)/abi.cpp $(wildcard $(CSRC)/*.hip)
OBJS  := $(patsubst $(CSRC)/%,build/%.o,$(SRCS))
LIB   := tmdiff_amd/libtmdiff_hip.so
HIPFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Iinclude -I$(CSRC) -Wall -Wno-unused-function

all: $(LIB)

build/%.o: $(CSRC)/% include/tmdiff_hip.h $(CSRC)/common.h $(CSRC)/epilogue.h $(CSRC)/bufaddr.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) $(OBJS) -o $@

clean:
	rm -rf build $(LIB)

.PHONY: all clean
