# Builds libmnt753_hip.so (HIP kernels + C ABI) for gfx950, the CPU oracle, and (when /root/reference is
# present) the reference build used to pin the oracle.  `make -j8` from the repo root.
PKG   := snark-challenge-prover-reference_amd
CSRC  := $(PKG)/csrc
BUILD := build
HIPCC ?= hipcc
HIPFLAGS := -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-result
LIB   := $(PKG)/libmnt753_hip.so
# test infrastructure beside the product (include/mnt753_hip_test.h): synthetic bases with known discrete logarithms, device-level
# known-answer hooks.  Links against the product library; nothing of the product links against it.
TESTLIB := $(PKG)/libmnt753_hip_test.so

HIP_SRCS := mnt753_core.hip mnt753_msm.hip msm_sort.hip msm_inst_mnt4g1.hip msm_inst_mnt4g2.hip msm_inst_mnt6g1.hip msm_inst_mnt6g2.hip mnt753_fft.hip mnt753_synth.hip mnt753_r1cs.hip mnt753_exchange.hip mnt753_selftest.hip
HIP_OBJS := $(addprefix $(BUILD)/,$(HIP_SRCS:.hip=.o))
TEST_SRCS := mnt753_testhooks.hip mnt753_synth_points.hip
TEST_OBJS := $(addprefix $(BUILD)/,$(TEST_SRCS:.hip=.o))
HDRS := $(wildcard $(CSRC)/*.hpp $(CSRC)/*.hip.h $(CSRC)/*.h include/*.h)

HOST := $(PKG)/host
MAIN := $(PKG)/main_hip

LAZY_TEST := $(PKG)/lazy_c_test

all: $(LIB) $(TESTLIB) $(MAIN) $(LAZY_TEST) oracle

# the wrapper's expression fusion from the caller's side (tools/host_tests/lazy_c_test.cpp; run by tests/test_prover_gpu.py)
$(LAZY_TEST): tools/host_tests/lazy_c_test.cpp $(HOST)/prover_hip_functions.cpp include/prover_hip_functions.hpp include/mnt753_hip.h $(LIB)
	g++ -O2 -std=c++17 -pthread -o $@ tools/host_tests/lazy_c_test.cpp $(HOST)/prover_hip_functions.cpp -L$(PKG) -lmnt753_hip -Wl,-rpath,'$$ORIGIN'

$(MAIN): $(HOST)/main.cpp $(HOST)/prover_hip_functions.cpp include/prover_hip_functions.hpp include/mnt753_hip.h $(LIB)
	g++ -O2 -std=c++17 -pthread -o $@ $(HOST)/main.cpp $(HOST)/prover_hip_functions.cpp -L$(PKG) -lmnt753_hip -Wl,-rpath,'$$ORIGIN'

# per-object dependency files (-MMD): a change to one header rebuilds the translation units that include it, not all of them
# (the point-kernel units take minutes each)
# (an object left over from before the .d files existed has none: it then depends on every header, as it used to)
.SECONDEXPANSION:
$(BUILD)/%.o: $(CSRC)/%.hip $$(if $$(wildcard $(BUILD)/$$*.d),,$$(HDRS))
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -MMD -MP -c $< -o $@
-include $(HIP_OBJS:.o=.d) $(TEST_OBJS:.o=.d)

# -soname: the test library (and anything else) names the product by its soname, so a development variant loaded from another
# directory (MNT753_LIB) satisfies it instead of a second copy of the product being mapped beside it
$(LIB): $(HIP_OBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -Wl,-soname,libmnt753_hip.so -o $@ $(HIP_OBJS) -ldl

$(TESTLIB): $(TEST_OBJS) $(LIB)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -Wl,-soname,libmnt753_hip_test.so -o $@ $(TEST_OBJS) -L$(PKG) -lmnt753_hip -Wl,-rpath,'$$ORIGIN'

oracle:
	$(MAKE) -C oracle

# Host-only sanitizer builds (no GPU needed, GPU sanitizers do not exist on the pool): main.cpp + the wrapper over a TEST STUB of the
# C ABI (tools/stub_abi: host memory, no arithmetic) under ASan + UBSan and under TSan.  tests/test_sanitizers_cpu.py runs them.
SAN_SRCS := $(HOST)/main.cpp $(HOST)/prover_hip_functions.cpp tools/stub_abi/stub_mnt753.cpp
asan: $(BUILD)/san/main_hip_asan
tsan: $(BUILD)/san/main_hip_tsan
$(BUILD)/san/main_hip_asan: $(SAN_SRCS) include/prover_hip_functions.hpp include/mnt753_hip.h
	@mkdir -p $(BUILD)/san
	g++ -O1 -g -std=c++17 -pthread -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -o $@ $(SAN_SRCS)
$(BUILD)/san/main_hip_tsan: $(SAN_SRCS) include/prover_hip_functions.hpp include/mnt753_hip.h
	@mkdir -p $(BUILD)/san
	g++ -O1 -g -std=c++17 -pthread -fsanitize=thread -o $@ $(SAN_SRCS)

clean:
	rm -rf $(BUILD) $(LIB) $(TESTLIB) $(MAIN) $(LAZY_TEST)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean asan tsan
