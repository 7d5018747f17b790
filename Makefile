# Build of the product library (HIP kernels + C ABI) and of the test oracle.
#   make            -> generalized_rbda_amd/libgrbda_hip.so, oracle/_build/libgrbda_oracle.so
#   make ref        -> oracle/_ref/libgrbda_codegen_ref.so (needs /root/reference; see oracle/Makefile)
HIPCC ?= hipcc
ARCH  ?= gfx950
CSRC  := generalized_rbda_amd/csrc
OBJ   := build/obj
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function
# the SLP vectoriser pairs fp32 operations into v_pk_* instructions and pays for it in register moves
# (measured: 6-10% slower f32 ABA); the kernels are scalar-per-lane code
KERNFLAGS := -fno-slp-vectorize

LIB := generalized_rbda_amd/libgrbda_hip.so

all: $(LIB) oracle

$(OBJ)/kernels.o: $(CSRC)/kernels.hip $(CSRC)/plan.h $(CSRC)/devplan.h $(CSRC)/devmath.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) $(KERNFLAGS) -c $< -o $@
# The chain kernels may re-associate sums (no signed zeros, no traps; NOT reciprocal / finite-math-only / approximate
# functions): lone multiplies and adds of the spatial products fold into FMAs, 10 % fewer floating-point instructions
# (MIT humanoid fp32 ABA 0.173 -> 0.170 ms, JVRC-1 and TelloWithArms 3-4 %); parity tolerances unchanged.
CHAINFLAGS := -fassociative-math -fno-signed-zeros -fno-trapping-math
$(OBJ)/chain_kernels.o: $(CSRC)/chain_kernels.hip $(CSRC)/gen_segments.h $(CSRC)/gen_rnea_segments.h $(CSRC)/plan.h $(CSRC)/devplan.h $(CSRC)/devmath.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) $(KERNFLAGS) $(CHAINFLAGS) -c $< -o $@
# second unit of the same source: the two fp64 kernels with the deepest register pressure (see the head of chain_kernels.hip)
$(OBJ)/chain_kernels_u1.o: $(CSRC)/chain_kernels.hip $(CSRC)/gen_segments.h $(CSRC)/gen_rnea_segments.h $(CSRC)/plan.h $(CSRC)/devplan.h $(CSRC)/devmath.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) $(KERNFLAGS) $(CHAINFLAGS) -DGRBDA_CHAIN_UNIT=1 -c $< -o $@
# third unit: the kernels of chain programs with generic clusters (gen_segments.h)
$(OBJ)/chain_kernels_u2.o: $(CSRC)/chain_kernels.hip $(CSRC)/gen_segments.h $(CSRC)/gen_rnea_segments.h $(CSRC)/plan.h $(CSRC)/devplan.h $(CSRC)/devmath.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) $(KERNFLAGS) $(CHAINFLAGS) -DGRBDA_CHAIN_UNIT=2 $(U2FLAGS) -c $< -o $@
# fourth unit: the latency-mode kernel with four wavefronts per tile (its [K | y0] blocks are LDS objects: GRBDA_KLDS)
$(OBJ)/chain_kernels_u3.o: $(CSRC)/chain_kernels.hip $(CSRC)/gen_segments.h $(CSRC)/gen_rnea_segments.h $(CSRC)/plan.h $(CSRC)/devplan.h $(CSRC)/devmath.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) $(KERNFLAGS) $(CHAINFLAGS) -DGRBDA_CHAIN_UNIT=3 -c $< -o $@
$(OBJ)/crba_kernels.o: $(CSRC)/crba_kernels.hip $(CSRC)/plan.h $(CSRC)/devplan.h $(CSRC)/devmath.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) $(KERNFLAGS) -c $< -o $@
$(OBJ)/deriv_kernels.o: $(CSRC)/deriv_kernels.hip $(CSRC)/plan.h $(CSRC)/devplan.h $(CSRC)/devmath.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) $(KERNFLAGS) -c $< -o $@
$(OBJ)/minv_kernels.o: $(CSRC)/minv_kernels.hip $(CSRC)/plan.h $(CSRC)/devplan.h $(CSRC)/devmath.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) $(KERNFLAGS) -c $< -o $@
$(OBJ)/manifold_kernels.o: $(CSRC)/manifold_kernels.hip $(CSRC)/plan.h $(CSRC)/devplan.h $(CSRC)/devmath.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) $(KERNFLAGS) -c $< -o $@
$(OBJ)/capi.o: $(CSRC)/capi.cpp $(CSRC)/plan.h $(CSRC)/devplan.h include/grbda_hip.h include/grbda_model_desc.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@
$(OBJ)/plan.o: $(CSRC)/plan.cpp $(CSRC)/plan.h include/grbda_hip.h include/grbda_model_desc.h
	@mkdir -p $(OBJ)
	g++ -O2 -std=c++17 -fPIC -Wall -c $< -o $@
$(OBJ)/urdf.o: $(CSRC)/urdf.cpp include/grbda_hip.h include/grbda_model_desc.h generalized_rbda_amd/include/grbda/ModelDescription.h
	@mkdir -p $(OBJ)
	g++ -O2 -std=c++17 -fPIC -Wall -c $< -o $@

# a build variant of the chain kernels for A/B runs (both units): make variant VARIANT=x VFLAGS="-D..." [U1FLAGS="..."]
# -> build/variants/libgrbda_hip_x.so (GRBDA_HIP_LIB selects it)
VARIANT ?= v
VFLAGS ?=
U1FLAGS ?=
U2FLAGS ?=
# (VFLAGS=-DGRBDA_EXP also compiles the ablation switches GRBDA_CHAIN_DEBUG / GRBDA_DEBUG_SWEEPS into the variant: the product
# library ignores them)
variant: $(OBJ)/chain_kernels_u2.o $(OBJ)/chain_kernels_u3.o $(OBJ)/kernels.o $(OBJ)/crba_kernels.o $(OBJ)/deriv_kernels.o $(OBJ)/minv_kernels.o $(OBJ)/manifold_kernels.o $(OBJ)/plan.o $(OBJ)/urdf.o
	@mkdir -p build/variants
	$(HIPCC) $(HIPFLAGS) $(VFLAGS) -x hip -c $(CSRC)/capi.cpp -o build/variants/capi_$(VARIANT).o
	$(HIPCC) $(HIPFLAGS) $(KERNFLAGS) $(CHAINFLAGS) $(VFLAGS) -c $(CSRC)/chain_kernels.hip -o build/variants/chain_kernels_$(VARIANT).o
	$(HIPCC) $(HIPFLAGS) $(KERNFLAGS) $(CHAINFLAGS) $(VFLAGS) -DGRBDA_CHAIN_UNIT=1 $(U1FLAGS) -c $(CSRC)/chain_kernels.hip -o build/variants/chain_kernels_u1_$(VARIANT).o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o build/variants/libgrbda_hip_$(VARIANT).so build/variants/chain_kernels_$(VARIANT).o build/variants/chain_kernels_u1_$(VARIANT).o build/variants/capi_$(VARIANT).o $^

$(LIB): $(OBJ)/kernels.o $(OBJ)/chain_kernels.o $(OBJ)/chain_kernels_u1.o $(OBJ)/chain_kernels_u2.o $(OBJ)/chain_kernels_u3.o $(OBJ)/crba_kernels.o $(OBJ)/deriv_kernels.o $(OBJ)/minv_kernels.o $(OBJ)/manifold_kernels.o $(OBJ)/capi.o $(OBJ)/plan.o $(OBJ)/urdf.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^

oracle:
	$(MAKE) -C oracle

ref:
	$(MAKE) -C oracle ref

clean:
	rm -rf build $(LIB) oracle/_build oracle/_ref

.PHONY: all oracle ref clean

# profiling variant with in-kernel cycle accounting (tools/prof_run.py); not part of `all`
prof: $(OBJ)/capi.o $(OBJ)/plan.o $(OBJ)/urdf.o $(OBJ)/chain_kernels.o $(OBJ)/chain_kernels_u1.o $(OBJ)/chain_kernels_u2.o $(OBJ)/crba_kernels.o $(OBJ)/deriv_kernels.o $(OBJ)/minv_kernels.o $(OBJ)/manifold_kernels.o
	@mkdir -p build/prof
	$(HIPCC) $(HIPFLAGS) $(KERNFLAGS) -DGRBDA_PROFILE -c $(CSRC)/kernels.hip -o build/prof/kernels.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o build/prof/libgrbda_hip_prof.so build/prof/kernels.o $^

# experiment builds: make exp NAME=foo DEFS="-DGRBDA_EXP_FOO" -> build/exp/libgrbda_foo.so
exp: $(OBJ)/capi.o $(OBJ)/plan.o $(OBJ)/urdf.o $(OBJ)/chain_kernels.o $(OBJ)/chain_kernels_u1.o $(OBJ)/chain_kernels_u2.o $(OBJ)/chain_kernels_u3.o $(OBJ)/crba_kernels.o $(OBJ)/deriv_kernels.o $(OBJ)/minv_kernels.o $(OBJ)/manifold_kernels.o
	@mkdir -p build/exp
	$(HIPCC) $(HIPFLAGS) $(KERNFLAGS) $(DEFS) -c $(CSRC)/kernels.hip -o build/exp/kernels_$(NAME).o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o build/exp/libgrbda_$(NAME).so build/exp/kernels_$(NAME).o $^

# experiment builds of the derivative kernels: make expd NAME=foo DEFS="-D..." -> build/exp/libgrbda_foo.so
expd: $(OBJ)/minv_kernels.o $(OBJ)/manifold_kernels.o $(OBJ)/capi.o $(OBJ)/plan.o $(OBJ)/urdf.o $(OBJ)/chain_kernels.o $(OBJ)/chain_kernels_u1.o $(OBJ)/chain_kernels_u2.o $(OBJ)/chain_kernels_u3.o $(OBJ)/crba_kernels.o $(OBJ)/kernels.o
	@mkdir -p build/exp
	$(HIPCC) $(HIPFLAGS) $(KERNFLAGS) $(DEFS) -c $(CSRC)/deriv_kernels.hip -o build/exp/deriv_$(NAME).o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o build/exp/libgrbda_$(NAME).so build/exp/deriv_$(NAME).o $^

# experiment builds of the chain kernels: make expc NAME=foo DEFS="-D..." -> build/exp/libgrbda_foo.so
expc: $(OBJ)/minv_kernels.o $(OBJ)/manifold_kernels.o $(OBJ)/capi.o $(OBJ)/plan.o $(OBJ)/urdf.o $(OBJ)/chain_kernels_u1.o $(OBJ)/chain_kernels_u2.o $(OBJ)/chain_kernels_u3.o $(OBJ)/deriv_kernels.o $(OBJ)/crba_kernels.o $(OBJ)/kernels.o
	@mkdir -p build/exp
	$(HIPCC) $(HIPFLAGS) $(KERNFLAGS) $(CHAINFLAGS) $(DEFS) -c $(CSRC)/chain_kernels.hip -o build/exp/chain_$(NAME).o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o build/exp/libgrbda_$(NAME).so build/exp/chain_$(NAME).o $^

# AddressSanitizer + UBSan build of the HOST-side code (CPU only: GPU sanitizers are not available on this pool): the plan
# compiler, the URDF+ reader and the oracle, driven by tools/asan_driver.cpp over every robot URDF and a serialised model.
#   make asan && build/asan/asan_driver tests/golden/robot-models/*.urdf
asan:
	@mkdir -p build/asan
	g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -Wall \
	    -DGRBDA_ASAN_DRIVER -I include -I generalized_rbda_amd/include tools/asan_driver.cpp $(CSRC)/plan.cpp $(CSRC)/urdf.cpp \
	    -x c oracle/grbda_oracle.c -x none -lm -lpthread -o build/asan/asan_driver
.PHONY: asan

# experiment builds of the manifold kernels: make expm NAME=foo DEFS="-D..." -> build/exp/libgrbda_foo.so
expm: $(OBJ)/minv_kernels.o $(OBJ)/deriv_kernels.o $(OBJ)/capi.o $(OBJ)/plan.o $(OBJ)/urdf.o $(OBJ)/chain_kernels.o $(OBJ)/chain_kernels_u1.o $(OBJ)/chain_kernels_u2.o $(OBJ)/chain_kernels_u3.o $(OBJ)/crba_kernels.o $(OBJ)/kernels.o
	@mkdir -p build/exp
	$(HIPCC) $(HIPFLAGS) $(KERNFLAGS) $(DEFS) -c $(CSRC)/manifold_kernels.hip -o build/exp/manifold_$(NAME).o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o build/exp/libgrbda_$(NAME).so build/exp/manifold_$(NAME).o $^

# experiment builds of the minv kernels: make expv NAME=foo DEFS="-D..." -> build/exp/libgrbda_foo.so
expv: $(OBJ)/deriv_kernels.o $(OBJ)/manifold_kernels.o $(OBJ)/capi.o $(OBJ)/plan.o $(OBJ)/urdf.o $(OBJ)/chain_kernels.o $(OBJ)/chain_kernels_u1.o $(OBJ)/chain_kernels_u2.o $(OBJ)/chain_kernels_u3.o $(OBJ)/crba_kernels.o $(OBJ)/kernels.o
	@mkdir -p build/exp
	$(HIPCC) $(HIPFLAGS) $(KERNFLAGS) $(DEFS) -c $(CSRC)/minv_kernels.hip -o build/exp/minv_$(NAME).o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o build/exp/libgrbda_$(NAME).so build/exp/minv_$(NAME).o $^
