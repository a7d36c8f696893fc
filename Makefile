# Builds libgsage_hip.so (the C-ABI hot-path library, gfx950 only) and the CPU oracle.
#   make            -> pytorch-graphsage_amd/libgsage_hip.so + oracle/libgsage_oracle.so
#   make hip        -> only the HIP library (hipcc cross-compiles without a GPU)
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
PKG := pytorch-graphsage_amd
SRC := $(wildcard $(PKG)/csrc/*.hip)
OBJ := $(SRC:.hip=.o)
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function -Wno-inline-asm

all: hip oracle

hip: $(PKG)/libgsage_hip.so

$(PKG)/csrc/%.o: $(PKG)/csrc/%.hip $(wildcard $(PKG)/csrc/*.h) include/gsage.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(PKG)/libgsage_hip.so: $(OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJ)

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(PKG)/csrc/*.o $(PKG)/libgsage_hip.so
	$(MAKE) -C oracle clean

.PHONY: all hip oracle clean
