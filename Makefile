# Build recipe for the MI355X-native detect engine (gfx950 only) and its test infrastructure.
#   make            -> libzly.so (HIP engine + C ABI), libzly_plugin.so (IInferenceEngine host side),
#                      host test binary, CPU oracle, synthetic weight files
#   make oracle     -> oracle/_build/libzly_oracle.so only (gcc, no GPU toolchain needed)
PKG      := zero-latency-yolo_amd
CSRC     := $(PKG)/csrc
HOST     := $(PKG)/host
OUT      := $(PKG)/_build
HIPCC    ?= hipcc
ARCH     ?= gfx950
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -I$(CSRC) -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-value
CXX      ?= g++
PY       ?= python3

KERNELS  := $(CSRC)/kernels_conv.hip $(CSRC)/kernels_pair.hip $(CSRC)/kernels_misc.hip $(CSRC)/kernels_head.hip $(CSRC)/kernels_stem.hip $(CSRC)/kernels_post.hip $(CSRC)/kernels_c2f64.hip $(CSRC)/kernels_sppf.hip
ENGINE   := $(CSRC)/engine.cpp $(CSRC)/weights.cpp
OBJS     := $(OUT)/kernels_conv.o $(OUT)/kernels_pair.o $(OUT)/kernels_misc.o $(OUT)/kernels_head.o $(OUT)/kernels_stem.o $(OUT)/kernels_post.o $(OUT)/kernels_c2f64.o $(OUT)/kernels_sppf.o $(OUT)/engine.o $(OUT)/weights.o

all: $(OUT)/libzly.so $(OUT)/libzly_gather.so $(OUT)/test_gather $(OUT)/zly_sharded_bench oracle weights host

$(OUT):
	mkdir -p $(OUT)

$(OUT)/%.o: $(CSRC)/%.hip $(CSRC)/zly_internal.h $(CSRC)/conv_device.h include/zly.h | $(OUT)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OUT)/engine.o: $(CSRC)/engine.cpp $(CSRC)/zly_internal.h $(CSRC)/weights.h include/zly.h | $(OUT)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(OUT)/weights.o: $(CSRC)/weights.cpp $(CSRC)/weights.h include/zly.h | $(OUT)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(OUT)/libzly.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

# ---- diagnostic build of the engine (never shipped, never loaded by tests / bench.py / the plugin): the same kernels, engine.cpp with -DZLY_DIAG, which
#      compiles in the result-changing ZLY_ABLATE_SKIP switch of tools/ablate_launches.sh (ZLY_LIB=.../libzly_diag.so python3 bench.py ...)
diaglib: $(OUT)/libzly_diag.so
$(OUT)/engine_diag.o: $(CSRC)/engine.cpp $(CSRC)/zly_internal.h $(CSRC)/weights.h include/zly.h | $(OUT)
	$(HIPCC) $(HIPFLAGS) -DZLY_DIAG=1 -x hip -c $< -o $@
$(OUT)/libzly_diag.so: $(filter-out $(OUT)/engine.o,$(OBJS)) $(OUT)/engine_diag.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^

# ---- in-process RCCL gather of result slabs (include/zly_gather.h): a library of its own -- it links RCCL, and a process that already carries
#      PyTorch's bundled RCCL (bench.py, the tests) must never load a second one
$(OUT)/libzly_gather.so: $(CSRC)/gather.cpp include/zly_gather.h | $(OUT)
	$(HIPCC) --offload-arch=$(ARCH) -O2 -std=c++17 -fPIC -shared -Iinclude -x hip $(CSRC)/gather.cpp -o $@ -L/opt/rocm/lib -lrccl

$(OUT)/test_gather: tests/cpp/test_gather.cpp $(OUT)/libzly_gather.so $(OUT)/libzly.so
	$(HIPCC) --offload-arch=$(ARCH) -O2 -std=c++17 -Wno-unused-value -Iinclude -x hip tests/cpp/test_gather.cpp -o $@ -L$(OUT) -lzly_gather -lzly -Wl,-rpath,'$$ORIGIN'

# the product caller of the gather library: one process, N GPUs in lock step (host/zly_sharded.hpp), as a bench / check tool
$(OUT)/zly_sharded_bench: $(PKG)/tools/bench_sharded.cpp $(HOST)/zly_sharded.hpp $(HOST)/zly_sharded_hip.hpp $(HOST)/zly_compat.hpp include/zly.h include/zly_gather.h $(OUT)/libzly_gather.so $(OUT)/libzly.so
	$(HIPCC) --offload-arch=$(ARCH) -O2 -std=c++17 -Wno-unused-value -Iinclude -I$(HOST) -x hip $(PKG)/tools/bench_sharded.cpp -o $@ -L$(OUT) -lzly_gather -lzly -Wl,-rpath,'$$ORIGIN'

# ... and its host logic against stubs of both C ABIs and of the device runtime (test infrastructure: no GPU, no libzly.so)
$(OUT)/test_sharded_stub: tests/cpp/test_sharded_stub.cpp $(HOST)/zly_sharded.hpp $(HOST)/zly_compat.hpp include/zly.h include/zly_gather.h | $(OUT)
	$(CXX) -O2 -std=c++17 -Wall -Iinclude -I$(HOST) -o $@ tests/cpp/test_sharded_stub.cpp

# ---- host side: the reference's IInferenceEngine plugin interface over the C ABI -----------------
host: $(OUT)/libzly_plugin.so $(OUT)/test_hip_engine $(OUT)/test_wire $(OUT)/test_game_step $(OUT)/test_frame_server $(OUT)/zly_h2h_bench $(OUT)/test_plugin_stub $(OUT)/test_sharded_stub

$(OUT)/libzly_plugin.so: $(HOST)/hip_inference_engine.cpp $(HOST)/hip_inference_engine.h $(HOST)/zly_sha256.hpp $(HOST)/zly_compat.hpp include/zly.h $(OUT)/libzly.so
	$(CXX) -O2 -std=c++17 -fPIC -shared -Iinclude -I$(HOST) -o $@ $(HOST)/hip_inference_engine.cpp -L$(OUT) -lzly -pthread -Wl,-rpath,'$$ORIGIN'

$(OUT)/test_hip_engine: tests/cpp/test_hip_engine.cpp $(OUT)/libzly_plugin.so
	$(CXX) -O2 -std=c++17 -Iinclude -I$(HOST) -o $@ tests/cpp/test_hip_engine.cpp -Wl,--no-as-needed -L$(OUT) -lzly_plugin -lzly -pthread -Wl,-rpath,'$$ORIGIN'

$(OUT)/zly_h2h_bench: $(PKG)/tools/bench_h2h.cpp $(OUT)/libzly_plugin.so
	$(CXX) -O2 -std=c++17 -Wall -Iinclude -I$(HOST) -o $@ $(PKG)/tools/bench_h2h.cpp -Wl,--no-as-needed -L$(OUT) -lzly_plugin -lzly -pthread -Wl,-rpath,'$$ORIGIN'

$(OUT)/test_frame_server: tests/cpp/test_frame_server.cpp $(HOST)/zly_frame_server.hpp $(HOST)/zly_wire.hpp $(HOST)/zly_game_step.hpp $(OUT)/libzly_plugin.so
	$(CXX) -O2 -std=c++17 -Wall -Iinclude -I$(HOST) -o $@ tests/cpp/test_frame_server.cpp -Wl,--no-as-needed -L$(OUT) -lzly_plugin -lzly -pthread -Wl,-rpath,'$$ORIGIN'

# the plugin's host logic against a link-time stub of the C ABI (test infrastructure: no libzly.so, no GPU)
$(OUT)/test_plugin_stub: tests/cpp/test_plugin_stub.cpp $(HOST)/hip_inference_engine.cpp $(HOST)/hip_inference_engine.h $(HOST)/zly_compat.hpp include/zly.h | $(OUT)
	$(CXX) -O2 -std=c++17 -Wall -Iinclude -I$(HOST) -o $@ tests/cpp/test_plugin_stub.cpp -pthread

$(OUT)/test_wire: tests/cpp/test_wire.cpp $(HOST)/zly_wire.hpp $(HOST)/zly_sha256.hpp $(HOST)/zly_compat.hpp | $(OUT)
	$(CXX) -O2 -std=c++17 -Wall -I$(HOST) -o $@ tests/cpp/test_wire.cpp

$(OUT)/test_game_step: tests/cpp/test_game_step.cpp $(HOST)/zly_game_step.hpp $(HOST)/zly_compat.hpp | $(OUT)
	$(CXX) -O2 -std=c++17 -Wall -I$(HOST) -o $@ tests/cpp/test_game_step.cpp

# ---- CPU oracle (test infrastructure only) --------------------------------------------------------
oracle: oracle/_build/libzly_oracle.so

oracle/_build/libzly_oracle.so: oracle/zly_oracle.c
	mkdir -p oracle/_build
	gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC -o $@ $<

# ---- seeded synthetic weights (no real weights exist offline) --------------------------------------
weights: $(OUT)/yolov8n_synth.zlyw

$(OUT)/yolov8n_synth.zlyw: $(PKG)/tools/zly_model.py | $(OUT)
	$(PY) $(PKG)/tools/zly_model.py --scale n -o $@

# ---- diagnostic build of the LDS conv kernel with s_memtime stamps per phase (never linked into libzly.so) --------
diag: $(OUT)/diag_lds

$(OUT)/diag_lds: $(PKG)/tools/diag_lds.hip $(CSRC)/kernels_conv.hip $(CSRC)/zly_internal.h | $(OUT)
	$(HIPCC) $(HIPFLAGS) -DZLY_DIAG=1 $< -o $@

clean:
	rm -rf $(OUT) oracle/_build

.PHONY: all host oracle weights diag diaglib clean
