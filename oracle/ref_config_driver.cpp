// Driver for the ONE translation unit of the reference that builds without third-party libraries:
// src/Config.cpp (flag parser).  Linked against /root/reference/src/Config.cpp where it lies (never copied);
// prints the parsed globals so that tests can pin the oracle's and the product's CLI parsers on the real thing.
#include <cstdio>
#include "Config.h"
int main(int argc, char* argv[]) {
    bool ok = parse_arguments(argc, argv);
    std::printf("ok=%d star=%d pt=%d pl=%d pm=%d mq=%d dp=%d di=%d w=%d r=%.17g a=%d b=%s c=%s f=%s o=%s G=%d CO=%d TO=%d RG=%d\n", (int)ok, (int)UsingSTAR, (int)Phred_Type,
                (int)Max_LowPhred_Len, (int)Min_Phred, (int)Min_MapQual, Concord_Dist_Pos, Concord_Dist_Idx, Min_Edge_Weight, DiscordantRatio, MaxAllowedDegree, Input_BAM.c_str(),
                Input_Chim_BAM.c_str(), Input_FASTA.c_str(), Output_Prefix.c_str(), (int)Print_Graph, (int)Print_Components_Ordering, (int)Print_Total_Ordering, (int)Print_Rearranged_Genome);
    return 0;
}
