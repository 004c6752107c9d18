# Source-compatibility check of the PRODUCT's headers (include/deciphon_host.h through the
# forwarding headers of include/compat/): the reference's own tests are compiled UNCHANGED, from
# where they lie under /root/reference, against this library.  This is not an oracle and nothing here
# computes an expected value -- the expected values are the ones written in those tests.
#   test/protein_model.c    model builder API; runs on the CPU (no Viterbi)
#   test/protein_profile.c  goldens G1-G3 (null / alt logliks, path ends, ten codons); its
#                           imm_dp_viterbi calls run on the MI355X
# (test/protein_db.c needs imm_example2_seq, test/protein_h3reader.c a network download,
#  test/standard_profile.c imm's generic HMM builder: not reachable)
# Outputs only into oracle/_ref/compat_tests/ (git-ignored; travels to the GPU box).  Reference
# sources are never copied: the .c files are read in place, and the binaries are not built at all
# where /root/reference does not exist.
REF     ?= /root/reference
ROOT    := $(abspath $(dir $(lastword $(MAKEFILE_LIST)))/..)
OUT     := $(ROOT)/oracle/_ref/compat_tests
CC      ?= gcc
CFLAGS  := -std=gnu11 -O1 -g -Wall -Werror=implicit-function-declaration -I$(ROOT)/include -I$(ROOT)/include/compat
LDFLAGS := -L$(ROOT)/deciphon-old_amd -ldeciphon_host -ldcp_hip -lm -Wl,-rpath,'$$ORIGIN/../../../deciphon-old_amd'
TESTS   := protein_model protein_profile

all: $(addprefix $(OUT)/,$(TESTS))

$(OUT)/%: $(REF)/test/%.c $(ROOT)/include/deciphon_host.h $(ROOT)/deciphon-old_amd/libdeciphon_host.so
	mkdir -p $(OUT)
	$(CC) $(CFLAGS) $< -o $@ $(LDFLAGS)

clean:
	rm -rf $(OUT)

.PHONY: all clean
