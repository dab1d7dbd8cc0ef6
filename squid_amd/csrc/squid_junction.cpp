// `squid_junction <BEDPEfile> <Input_Chim_BAM> <FA_genome> <OUTPrefix>`: the command line of utils/JunctionSequence.cpp (:527-557) over
// sq_junction_sequences of libsquid_hip.so (host work only).
#include <cstdio>

#include "../../include/squid_hip.h"

int main(int argc, char** argv) {
    if (argc < 5) { std::printf("junctionsequence <BEDPEfile> <Input_Chim_BAM> <FA_genome> <OUTPrefix>\n"); return 0; }
    char err[512] = {0};
    const int rc = sq_junction_sequences(argv[1], argv[2], argv[3], argv[4], err, sizeof err);
    if (rc) { std::fprintf(stderr, "squid_junction: %s (%s)\n", sq_strerror(rc), err); return 1; }
    return 0;
}
