# oracle/ref.mk -- builds the REAL reference (tangkk/audiomod phase-vocoder path)
# from its own sources where they lie under $(REF) into oracle/_ref/ (git-ignored).
#
# TEST INFRASTRUCTURE ONLY.  Nothing here is shipped or linked into the product
# library.  No reference source is copied into this repository: the compiler reads the
# files in place.  The reference's own build system (CMake) is NOT run; this is a
# short hand-written recipe with the same effective flags the reference's Release
# build uses (-O3 -DNDEBUG -std=gnu++14, baseline x86-64 => no FMA contraction).
#
#   make -f oracle/ref.mk            (from the repo root; needs /root/reference)
#
REF ?= /root/reference
OUT := oracle/_ref
CXX := g++
CC  := gcc
CXXFLAGS := -O3 -DNDEBUG -std=gnu++14 -fPIC -w
CFLAGS   := -O3 -DNDEBUG -fPIC -w
INC := -I$(REF)/src -I$(REF)/include -I$(REF)/include/dafx -I$(REF)/include/analyzer

# Only the files the phase-vocoder path needs (SURVEY.md section 2, rows 1-13 + 16).
CXXSRC := $(wildcard $(REF)/src/phasevocoder/*.cc) \
          $(REF)/src/common/dsp/FFT.cc $(REF)/src/common/dsp/resampler.cc \
          $(REF)/src/common/system/sys.cc \
          $(wildcard $(REF)/src/common/gen/*.cc)
CSRC   := $(wildcard $(REF)/src/common/kissfft/*.c) $(REF)/src/common/speex/resample.c

CXXOBJ := $(patsubst $(REF)/%.cc,$(OUT)/obj/%.o,$(CXXSRC))
COBJ   := $(patsubst $(REF)/%.c,$(OUT)/obj/%.o,$(CSRC))

all: $(OUT)/ref_driver $(OUT)/ref_kat $(OUT)/ref_formant $(OUT)/audiomod-exe

$(OUT)/obj/%.o: $(REF)/%.cc
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS) $(INC) -c $< -o $@

$(OUT)/obj/%.o: $(REF)/%.c
	@mkdir -p $(dir $@)
	$(CC) $(CFLAGS) $(INC) -c $< -o $@

$(OUT)/libaudiomod_ref.a: $(CXXOBJ) $(COBJ)
	ar rcs $@ $^

# ref_driver.cc / ref_kat.cc are OUR code: thin callers of the reference's public API.
$(OUT)/ref_driver: oracle/ref_driver.cc $(OUT)/libaudiomod_ref.a
	$(CXX) $(CXXFLAGS) $(INC) $< $(OUT)/libaudiomod_ref.a -o $@ -static-libstdc++ -static-libgcc -lpthread -ldl

$(OUT)/ref_kat: oracle/ref_kat.cc $(OUT)/libaudiomod_ref.a
	$(CXX) $(CXXFLAGS) $(INC) $< $(OUT)/libaudiomod_ref.a -o $@ -static-libstdc++ -static-libgcc -lpthread -ldl

# the reference's (dead-code) cepstral formant shift, called directly
$(OUT)/ref_formant: oracle/ref_formant.cc $(OUT)/libaudiomod_ref.a
	$(CXX) $(CXXFLAGS) $(INC) $< $(OUT)/libaudiomod_ref.a -o $@ -static-libstdc++ -static-libgcc -lpthread -ldl

# The reference's own CLI (used only to capture golden WAV files for the CLI/WAV counterpart).
# main.cc instantiates every effect of the library, so this target compiles the whole reference tree.
ALLCXX := $(wildcard $(REF)/src/*/*.cc) $(wildcard $(REF)/src/common/*/*.cc) $(wildcard $(REF)/main/*.cc)
ALLOBJ := $(patsubst $(REF)/%.cc,$(OUT)/obj/%.o,$(ALLCXX))
$(OUT)/audiomod-exe: $(ALLOBJ) $(COBJ)
	$(CXX) $(CXXFLAGS) $(ALLOBJ) $(COBJ) -o $@ -static-libstdc++ -static-libgcc -lpthread -ldl

# ---------------------------------------------------------------------------------------------------------
# "Drops into audiomod-exe unchanged", as a build: the reference's own main/main.cc + wavfile.cc and every
# non-phase-vocoder effect, compiled from where they lie, linked against THIS repository's audiomod::phasevocoder
# (audiomod_amd/csrc/phasevocoder.cc over libaudiomod_pv.so) instead of src/phasevocoder/*.  What a maintainer
# changes is one header: include/dafx/phasevocoder.h.  The reference headers include each other by quoted paths
# relative to themselves, so that swap is materialised as a tree of symlinks ($(OUT)/dropin/include: every
# reference header except that one, which points at ours) -- no reference file is copied or edited.
# tests/test_dropin_exe.py runs the result on the GPU box against the reference CLI's golden WAV files.
# ---------------------------------------------------------------------------------------------------------
DROPIN := $(OUT)/dropin
REPO   := $(abspath .)
NONPV_OBJ := $(filter-out $(OUT)/obj/src/phasevocoder/%,$(filter-out $(OUT)/obj/main/%,$(ALLOBJ)))
$(DROPIN)/include/.stamp: include/dafx/phasevocoder.h
	@rm -rf $(DROPIN)/include && mkdir -p $(DROPIN)/include/dafx $(DROPIN)/include/analyzer
	@for f in $(REF)/include/*.h; do ln -s $$f $(DROPIN)/include/; done
	@for f in $(REF)/include/analyzer/*.h; do ln -s $$f $(DROPIN)/include/analyzer/; done
	@for f in $(REF)/include/dafx/*.h; do [ "$$(basename $$f)" = phasevocoder.h ] || ln -s $$f $(DROPIN)/include/dafx/; done
	@ln -s $(REPO)/include/dafx/phasevocoder.h $(DROPIN)/include/dafx/phasevocoder.h
	@touch $@
DROPIN_INC := -I$(DROPIN)/include -I$(DROPIN)/include/dafx -I$(DROPIN)/include/analyzer -I$(REF)/src
$(DROPIN)/main.o: $(REF)/main/main.cc $(DROPIN)/include/.stamp
	$(CXX) $(CXXFLAGS) $(DROPIN_INC) -c $< -o $@
$(DROPIN)/wavfile.o: $(REF)/main/wavfile.cc $(DROPIN)/include/.stamp
	$(CXX) $(CXXFLAGS) $(DROPIN_INC) -c $< -o $@
$(DROPIN)/phasevocoder_shim.o: audiomod_amd/csrc/phasevocoder.cc $(DROPIN)/include/.stamp include/audiomod_pv.h
	$(CXX) $(CXXFLAGS) -std=gnu++17 $(DROPIN_INC) -Iinclude -c $< -o $@
$(OUT)/audiomod-exe-mi355x: $(DROPIN)/main.o $(DROPIN)/wavfile.o $(DROPIN)/phasevocoder_shim.o $(NONPV_OBJ) $(COBJ) audiomod_amd/lib/libaudiomod_pv.so
	$(CXX) $(CXXFLAGS) $(DROPIN)/main.o $(DROPIN)/wavfile.o $(DROPIN)/phasevocoder_shim.o $(NONPV_OBJ) $(COBJ) \
	    -Laudiomod_amd/lib -laudiomod_pv -Wl,-rpath,'$$ORIGIN/../../audiomod_amd/lib' -Wl,-rpath-link,/opt/rocm/lib \
	    -o $@ -lpthread -ldl
dropin: $(OUT)/audiomod-exe-mi355x
all: dropin

clean:
	rm -rf $(OUT)
.PHONY: all clean dropin
