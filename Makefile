# Top-level convenience targets.
#   make            : the product library (audio_codec_amd/liblc3plus_hip.so) + the WAV encoder / decoder front ends
#   make relink     : the reference's unmodified CLI (codec_exe.c) linked against the HIP engine -> oracle/_ref/LC3plus_hip
#   make reference  : what the reference repo's (broken) `make reference` was meant to give: the ETSI float tools built from
#                     /root/reference into oracle/_ref/ (test infrastructure) next to our front end linked against the HIP engine
#   make test       : CPU test-suite
HIPCC ?= hipcc
CC    ?= gcc
all: lib cli
lib:
	$(MAKE) -s -C audio_codec_amd/csrc
cli: lib tools/lc3plus_enc_cli tools/lc3plus_dec_cli
tools/lc3plus_enc_cli: tools/lc3plus_enc_cli.c include/lc3.h include/lc3plus_batch.h audio_codec_amd/liblc3plus_hip.so
	$(CC) -std=c99 -O2 -Wall -Iinclude -o $@ tools/lc3plus_enc_cli.c -Laudio_codec_amd -llc3plus_hip -Wl,-rpath,'$$ORIGIN/../audio_codec_amd'
tools/lc3plus_dec_cli: tools/lc3plus_dec_cli.c include/lc3.h include/lc3plus_batch.h audio_codec_amd/liblc3plus_hip.so
	$(CC) -std=c99 -O2 -Wall -Iinclude -o $@ tools/lc3plus_dec_cli.c -Laudio_codec_amd -llc3plus_hip -Wl,-rpath,'$$ORIGIN/../audio_codec_amd'
# The reference's own command-line tool relinked against this library: R/codec_exe.c compiled WHERE IT LIES against the reference's
# headers, every codec object of the reference replaced by liblc3plus_hip.so (SURVEY 8b "Who calls it").  Output under oracle/_ref/
# (git-ignored, travels with gpurun snapshots).  Only possible where /root/reference is mounted.
REF_FL ?= /root/reference/LC3plus_ETSI_src_v17171_20200723/src/floating_point
relink: lib
	@if [ -f $(REF_FL)/codec_exe.c ]; then mkdir -p oracle/_ref && \
	  $(CC) -std=c99 -O2 -w -I$(REF_FL) -o oracle/_ref/LC3plus_hip $(REF_FL)/codec_exe.c -Laudio_codec_amd -llc3plus_hip -lm -Wl,-rpath,'$$ORIGIN/../../audio_codec_amd' ; \
	else echo "Makefile: $(REF_FL) not present -- keeping prebuilt oracle/_ref/LC3plus_hip (if any)"; fi
reference: all relink
	$(MAKE) -s -C oracle ref restatement
test:
	python -m pytest tests -x -q -m "not gpu"
clean:
	$(MAKE) -s -C audio_codec_amd/csrc clean
	$(MAKE) -s -C oracle clean
	rm -f tools/lc3plus_enc_cli tools/lc3plus_dec_cli
.PHONY: all lib cli reference relink test clean
