# Top-level convenience targets.
#   make            : the product library (audio_codec_amd/liblc3plus_hip.so) + the WAV encoder / decoder front ends
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
reference: all
	$(MAKE) -s -C oracle ref restatement
test:
	python -m pytest tests -x -q -m "not gpu"
clean:
	$(MAKE) -s -C audio_codec_amd/csrc clean
	$(MAKE) -s -C oracle clean
	rm -f tools/lc3plus_enc_cli tools/lc3plus_dec_cli
.PHONY: all lib cli reference test clean
