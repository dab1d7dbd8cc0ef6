/*
 * squid_hip.h -- C ABI of the MI355X-native SQUID hot path (libsquid_hip.so).
 *
 * The reference (Kingsford-Group/squid v1.5) has no plugin / FFI boundary: its hot path sits behind C++
 * signatures inside one process.  Each entry point below names the reference
 * interface it replaces (file:line under the reference's src/).  Conventions:
 *   - every function returns 0 on success or a negative SQ_E_* code; sq_strerror() names it;
 *   - no C++ types or exceptions cross the boundary; inputs are caller-owned and only read during the call;
 *   - outputs are library-owned views, valid until the next call on the same context or sq_destroy();
 *   - one context <-> one GPU <-> one host thread at a time (thread-compatible, not thread-safe);
 *   - the library refuses to work without a HIP device: there is no CPU code path for the GPU stages.
 */
#ifndef SQUID_HIP_H
#define SQUID_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SQ_ABI_VERSION 5

enum {
    SQ_OK = 0,
    SQ_E_ARG = -1,       /* bad argument / call order */
    SQ_E_NODEVICE = -2,  /* no usable HIP device (the product never falls back to the CPU) */
    SQ_E_HIP = -3,       /* a HIP runtime call failed */
    SQ_E_IO = -4,        /* cannot open / parse a BAM file */
    SQ_E_UNSORTED = -5,  /* concordant input is not coordinate sorted (README.md:23 requires it) */
    SQ_E_ASSERT = -6,    /* the reference would hit one of its live assert()s on this input */
    SQ_E_CAPACITY = -7,  /* an internal table overflowed */
    SQ_E_EMPTYCHIM = -8, /* chimeric input has no usable record (reference: out-of-bounds read, ReadRec.cpp:379) */
    SQ_NEED_EXCHANGE = 1 /* not an error: a chromosome-sharded run needs an all-gather before it can continue (see below) */
};

/* The tuning globals of src/Config.h:23-49 (defaults = src/Config.cpp:11-37), plus device selection. */
typedef struct sq_params {
    int32_t abi_version;       /* = SQ_ABI_VERSION */
    int32_t device;            /* HIP device ordinal */
    int32_t phred_type;        /* Phred_Type: 1 => offset 33 (Config.cpp:19) */
    int32_t max_lowphred_len;  /* Max_LowPhred_Len (10) */
    int32_t min_phred;         /* Min_Phred (4) */
    int32_t min_mapqual;       /* Min_MapQual; the CLI applies the STAR => 255 rule (Config.cpp:221-222) */
    int32_t concord_dist_pos;  /* Concord_Dist_Pos (50000) */
    int32_t concord_dist_idx;  /* Concord_Dist_Idx (20) */
    int32_t min_edge_weight;   /* Min_Edge_Weight (5) */
    double discordant_ratio;   /* DiscordantRatio (8) */
    int32_t max_allowed_degree; /* MaxAllowedDegree (5) */
    int32_t rank, world_size;  /* chromosome sharding (sq_set_shard, sq_exchange_*); 0,1 for a single GPU */
} sq_params;

void sq_default_params(sq_params* p);

typedef struct sq_ctx sq_ctx;

/* SoA batch of decoded alignment records (what BamTools hands the reference one BamAlignment at a time,
 * src/ReadRec.cpp:10-88; SURVEY.md section 8(d) record layout).  Blocks are the aligned blocks that
 * ReadRec_t::ReadRec_t keeps, in CIGAR order, with ReadPos already strand-mirrored (ReadRec.cpp:74-75). */
typedef struct sq_aln_batch {
    int64_t n_rec, n_blk;
    const int32_t* refid;      /* BamAlignment::RefID */
    const int32_t* pos;        /* Position */
    const int32_t* mate_refid; /* MateRefID */
    const int32_t* mate_pos;   /* MatePosition */
    const int32_t* end_pos;    /* GetEndPosition() */
    const uint16_t* flag;      /* SAM flag */
    const uint8_t* mapq;       /* MapQuality */
    const uint8_t* aux;        /* SQ_AUX_* */
    const uint16_t* totlen;    /* TotalLen of ReadRec.cpp:16-18 */
    const uint32_t* blk_off;   /* n_rec+1 offsets into the block arrays */
    const int32_t* b_refpos;
    const int32_t* b_matchref;
    const uint16_t* b_readpos;
    const uint16_t* b_matchread;
    /* QNAMEs, only read by sq_ingest_chimeric (ReadRec.cpp:11-13,354): n_rec+1 offsets into name_blob */
    const uint32_t* name_off;
    const char* name_blob;
} sq_aln_batch;

#define SQ_AUX_MULTI 0x01    /* HasTag("XA") || IH > 1   (SegmentGraph.cpp:297-302) */
#define SQ_AUX_INCHIM 0x02   /* raw QNAME is in the chimeric name set (SegmentGraph.cpp:302), see sq_chim_contains */
#define SQ_AUX_LOWPHRED 0x04 /* low-Phred run longer than Max_LowPhred_Len (ReadRec.cpp:19-44) */

int sq_create(const sq_params* p, sq_ctx** out);
void sq_destroy(sq_ctx* c);
const char* sq_strerror(int code);
const char* sq_last_error(sq_ctx* c); /* free-text detail of the last failure on this context */

/* BuildRefName (src/ReadRec.cpp:267-283): reference lengths in header order. */
int sq_set_references(sq_ctx* c, int32_t n_ref, const int32_t* ref_len);

/* BuildChimericSBamRecord (src/ReadRec.cpp:329-413): all records of the chimeric BAM in file order; one call. */
int sq_ingest_chimeric(sq_ctx* c, const sq_aln_batch* b);
/* binary_search(ChimName, record.Name) of src/SegmentGraph.cpp:302,1584,3136 (incl. the "" entry, ledger B9) */
int sq_chim_contains(sq_ctx* c, const char* name, size_t len);
/* The concordant stream, in file order; may be called repeatedly (batches are appended in HBM). */
int sq_ingest_concordant(sq_ctx* c, const sq_aln_batch* b);

/* K0: the inflated BAM record stream itself (no header), `rec_off[i]` = byte offset of record i's block_size field.
 * The records are parsed on the GPU (k_parse_*): field decode + ReadRec_t::ReadRec_t (src/ReadRec.cpp:10-88) +
 * tag / QNAME tests (src/SegmentGraph.cpp:297-302).  Needs sq_ingest_chimeric first (QNAME set, quality flags). */
int sq_ingest_concordant_bam(sq_ctx* c, const uint8_t* bam, size_t nbytes, const uint64_t* rec_off, int64_t n_rec);

/* Convenience host-side readers (own BGZF/BAM decoder; replaces the BamTools calls listed in SURVEY.md
 * appendix C).  The chimeric reader decodes on the host; the concordant reader inflates and finds record boundaries on
 * the host and parses on the GPU (sq_ingest_concordant_bam), or decodes on host threads when SQUID_HOST_PARSE is set. */
int sq_read_header(const char* bam_path, int32_t* n_ref, int32_t* ref_len, char* names, size_t names_cap);
int sq_ingest_chimeric_file(sq_ctx* c, const char* bam_path);
int sq_ingest_concordant_file(sq_ctx* c, const char* bam_path, int32_t n_threads);
/* Both files in one call, the chimeric BAM decoded on a host thread while the GPU reader works on the concordant BAM (the
 * reference reads them one after the other: src/main.cpp:33-36 and src/SegmentGraph.cpp:293).  Same state afterwards as
 * sq_ingest_chimeric_file followed by sq_ingest_concordant_file; an error of either file is returned. */
int sq_ingest_files(sq_ctx* c, const char* chim_bam_path, const char* bam_path, int32_t n_threads);
/* `squid --bwa` (src/Config.cpp:98-100, src/main.cpp:33-37): ONE coordinate-sorted BAM in which split reads are supplementary records
 * and no separate chimeric file exists.  The records are decoded on host threads together with their QNAMEs and stay on the host;
 * sq_build_graph then runs BuildNode_BWA (src/SegmentGraph.cpp:833-1205) and RawEdges (:1698-1930, which also rebuilds the chimeric
 * fragments from the partially aligned reads) over them and continues with the same graph stages as the STAR path; sq_call_sv counts
 * the breakpoint support over the same batch.  sq_params.min_mapqual is the caller's (-mq, default 1: the STAR => 255 rule of
 * Config.cpp:221-222 does not apply).  A chimeric file ingested before (`-c` next to `--bwa`) only contributes its ReadLen, as in
 * the reference.  Not available to a chromosome-sharded context. */
int sq_ingest_bwa_file(sq_ctx* c, const char* bam_path, int32_t n_threads);
/* Benchmarks and repeated runs: sq_stage_bam copies the compressed bytes of a BAM file into HBM once; a later
 * sq_ingest_concordant_file on the same path then takes the GPU reader (BGZF inflate, record boundaries and record parse
 * on the device) straight from that copy, with no host->device transfer of the file (the host still walks the BGZF block
 * headers of the mapped file).  sq_clear_records drops the resident concordant records and every graph result but keeps
 * the device buffers, so the same context can ingest again (replaces destroying and re-creating the context). */
int sq_stage_bam(sq_ctx* c, const char* bam_path);
int sq_clear_records(sq_ctx* c);
/* On-disk cache of the resident concordant records: parameter sweeps over -w/-r/-a/-dp/-di/-mq
 * re-run sq_build_graph..sq_call_sv on the same records and can skip the BAM decode (the three BamReader passes of
 * SegmentGraph.cpp:260-347, :1553-1621, :3098-3178).  sq_save_records writes what the ingest calls have made resident;
 * sq_load_records replaces sq_ingest_concordant_file (call it after sq_set_references and sq_ingest_chimeric*: the
 * records carry the "QNAME is in the chimeric BAM" bit).  The file records the parse parameters (-pt/-pl/-pm), the
 * number of references and a hash of the chimeric name set; sq_load_records returns SQ_E_ARG when they differ from
 * the context's.  A chromosome-sharded context keeps only the records of its shard. */
/* The cache is also bound to the concordant BAM it was decoded from (size and modification time): sq_ingest_concordant_file
 * records them by itself; before sq_load_records the caller names the BAM with sq_set_source, and a cache written from
 * another (or a re-aligned) file is refused with SQ_E_ARG.  sq_load_records also refuses a context that already holds
 * concordant records. */
int sq_set_source(sq_ctx* c, const char* bam_path);
int sq_save_records(sq_ctx* c, const char* cache_path);
int sq_load_records(sq_ctx* c, const char* cache_path);

/* SegmentGraph_t::SegmentGraph_t(RefLength, Chimrecord, bam) -- src/SegmentGraph.cpp:104-124 */
int sq_build_graph(sq_ctx* c);

typedef struct sq_graph {
    int32_t n_nodes, n_edges;
    const int32_t *chr, *pos, *len, *support, *label;  /* Node_t (src/BPNode.h:26-57) + Label */
    const double* avgdepth;
    const int32_t *ind1, *ind2, *weight, *groupweight;  /* Edge_t (src/BPEdge.h:24-77) */
    const uint8_t *head1, *head2;
} sq_graph;
/* stage: 0 = final graph (what OutputGraph prints, src/SegmentGraph.cpp:3223-3234);
 *        1 = after BuildNode_STAR, 2 = after BuildEdges, 3 = after FilterbyWeight, 4 = after FilterEdges,
 *        5 = after CompressNode  (intermediate snapshots are kept for the parity tests) */
int sq_graph_view(sq_ctx* c, int32_t stage, sq_graph* g);
/* on = 0: sq_build_graph no longer keeps the flat copies of the INTERMEDIATE graphs (stages 1-5 above; five copies of up to millions of
 * nodes and edges on a dense sample, made for inspection and the parity tests -- the result does not need them); sq_graph_view then only
 * answers for stage 0.  Default: kept.  `build/squid` and bench.py switch them off. */
int sq_keep_stage_graphs(sq_ctx* c, int32_t on);

/* vector<vector<int>> Ordering() -- src/SegmentGraph.cpp:3236-3262: CSR of signed 1-based node ids */
typedef struct sq_orders {
    int32_t n_components;
    const int32_t* comp_off; /* n_components+1 */
    const int32_t* nodes;
} sq_orders;
int sq_order(sq_ctx* c, sq_orders* o);
/* The per-component orders stitched into whole new chromosomes: SortComponents -> MergeSingleton -> SortComponents -> MergeComponents
 * (src/main.cpp:45-48, src/SegmentGraph.cpp:4010-4504) -- what `-TO 1` prints as <prefix>_component.txt and `-RG 1` spells out as
 * <prefix>_genome.fa.  The SV calls do not depend on it; only callers that want those two outputs need it. */
int sq_total_order(sq_ctx* c, sq_orders* o);

/* ExactBreakpoint + ExactBPConcordantSupport + DeMultiplyDisEdges + the row selection of WriteBEDPE
 * (src/SegmentGraph.cpp:3019-3221,3012-3017; src/WriteIO.cpp:45-124).  One row per printed _sv.txt line. */
typedef struct sq_sv_table {
    int32_t n_rows;
    const int32_t *chr1, *start1, *end1, *chr2, *start2, *end2, *score, *sup1, *sup2;
    const uint8_t *strand1_minus, *strand2_minus;
} sq_sv_table;
int sq_call_sv(sq_ctx* c, sq_sv_table* t);

/* per-edge breakpoint table of the final graph (parity tests): CSR over edges in final sorted order */
typedef struct sq_bp_table {
    int32_t n_edges;
    const int32_t* bp_off;     /* n_edges+1 */
    const int32_t *bp1, *bp2;  /* -1,-1 when the edge has no split-read breakpoint */
    const int32_t *sup1, *sup2;
} sq_bp_table;
int sq_breakpoints(sq_ctx* c, sq_bp_table* t);

/* Multi-GPU: one context per rank (sq_params.rank / world_size); rank r holds the concordant
 * records of the RefIDs [first_ref, end_ref) -- contiguous ranges in rank order that cover all references -- and
 * every rank ingests the whole chimeric BAM (small).  Call sq_set_shard after sq_set_references and before the
 * concordant ingest; the ingest functions then keep only the records this rank owns.
 * sq_build_graph and sq_call_sv return SQ_NEED_EXCHANGE whenever they need data from the other shards: the caller
 * gets this rank's bytes with sq_exchange_pack, all-gathers them (variable length: RCCL / gloo through
 * torch.distributed, MPI, or a loop over in-process contexts), hands the concatenation back with
 * sq_exchange_unpack and calls the same function again, until it returns SQ_OK or an error.  All ranks end with
 * identical graphs, orders and SV tables.  The exchanges of sq_build_graph: three of a few bytes (stream boundaries
 * and seed nodes) and the data exchange proper -- per-node depth sums of the own chromosomes plus the locally reduced
 * concordant edges, which closes the inter-chromosomal edges; sq_call_sv: the per-breakpoint coverage counts. */
int sq_set_shard(sq_ctx* c, int32_t first_ref, int32_t end_ref);
int sq_exchange_pack(sq_ctx* c, const void** buf, int64_t* nbytes);
int sq_exchange_unpack(sq_ctx* c, const void* gathered, const int64_t* nbytes_per_rank, int32_t world_size);
/* The same exchange carried out by the library (SURVEY.md section 8(b) `sq_exchange(ctx, comm)`, 8(e) "one RCCL all-gather over xGMI"):
 * after SQ_NEED_EXCHANGE call sq_exchange, then the stage function again -- no pack / unpack in the caller.  The transport is installed
 * once per context:
 *   sq_rccl_init    rank 0 makes an id with sq_rccl_unique_id (128 bytes) and gets it to the other ranks by any means; every rank
 *                   passes it in and joins a communicator of sq_params.world_size ranks as sq_params.rank (ncclCommInitRank);
 *   sq_rccl_attach  a communicator (ncclComm_t) the caller already has; it stays the caller's;
 *   sq_set_allgather any fixed-size all-gather of host buffers: fn(user, send, nbytes, recv) fills recv with the world_size pieces
 *                   in rank order and returns 0 (MPI_Allgather, a gloo shim in the tests).
 * An exchange is ONE all-gather of 16 KiB pieces (length + payload; stream boundaries, seed nodes and breakpoint counts fit); only a
 * payload beyond that -- the node sums and reduced edges of a large graph -- takes a second one for the remainders. */
typedef int (*sq_allgather_fn)(void* user, const void* send, int64_t nbytes, void* recv);
int sq_set_allgather(sq_ctx* c, sq_allgather_fn fn, void* user);
int sq_rccl_unique_id(void* id128);
int sq_rccl_init(sq_ctx* c, const void* id128);
int sq_rccl_attach(sq_ctx* c, void* nccl_comm);
/* 1 when librccl can be bound in this process (dlopen; no communicator is made, no collective entered): the ranks of a sharded run agree
 * on this BEFORE any of them enters the collective ncclCommInitRank of sq_rccl_init.  sq_rccl_release drops the context's RCCL
 * transport (and a communicator the library made itself) when the caller falls back to sq_set_allgather. */
int sq_rccl_available(void);
int sq_rccl_release(sq_ctx* c);
int sq_exchange(sq_ctx* c);
int sq_exchange_stats(sq_ctx* c, int64_t* collectives, int64_t* bytes); /* all-gathers issued by sq_exchange so far, payload bytes received */

/* utils/JunctionSequence.cpp (`junctionsequence <BEDPE> <Chim_BAM> <FA_genome> <OUTPrefix>`, :527-557): the junction sequences of the SV
 * calls of a `_sv.txt` -- <prefix>_junc_precise.fa (calls narrowed to the bases split reads cover, with their split-read support),
 * _junc_relax.fa (every call; exact ones widened by 1000 bases) and _junc_alt.fa (alternative junction points within 5 bases).  A
 * consumer of the hot path's output: host work only, no context and no device needed.  On failure `errbuf` gets the detail. */
int sq_junction_sequences(const char* bedpe_path, const char* chim_bam_path, const char* fasta_path, const char* out_prefix, char* errbuf, size_t errcap);

/* Timing of the last sq_build_graph/sq_order/sq_call_sv on this context, measured with HIP events on the
 * library's own stream.  names[i] is a static string; ms[i] the accumulated duration; launches[i] the count. */
typedef struct sq_timing {
    int32_t n;
    const char* const* names;
    const double* ms;
    const int64_t* launches;
    const double* bytes; /* algorithmic bytes moved by the kernel (0 for host stages) */
    const double* busy_ms; /* kernels: time during which at least one launch of that name was running (launches of one name overlap
                              when they sit on several streams -- the BGZF reader --; ms[i] is the SUM of their durations) */
} sq_timing;
int sq_get_timing(sq_ctx* c, sq_timing* t);
/* keep = 1: sq_build_graph no longer clears the timing table, so it accumulates over repeated runs (bench loops read it once) */
int sq_timing_accumulate(sq_ctx* c, int32_t keep);
/* forget the process-wide mapping and BGZF block index of the last BAM file read: the next sq_ingest_*_file maps the file and walks its
 * block headers again (measurements of a first read; the cache only saves time, never changes results) */
int sq_drop_file_cache(void);
int sq_reset(sq_ctx* c); /* drop graph results, keep ingested records resident in HBM (bench re-runs) */
/* give back the device memory the GPU reader keeps between ingests -- the compressed bytes of the file range last streamed to HBM
 * (file-sized: 5.9 GB for C3), the staged copy of sq_stage_bam, the token / inflate buffers of the batch pipeline (~20 GB) -- and its
 * page-locked host buffers.  The resident records, the graph and every result stay.  The next ingest allocates again (what a first
 * ingest does anyway); a process that keeps a context for parameter sweeps over resident records calls this once after the ingest. */
int sq_release_reader_buffers(sq_ctx* c);
/* For a host program that runs sample after sample in ONE process (a service, bench.py): tell the C library's allocator to keep the
 * memory the host stages free instead of handing it back to the kernel after every sample (glibc: mallopt M_TRIM_THRESHOLD, M_TOP_PAD,
 * M_MMAP_THRESHOLD) -- the pairing of the chimeric records and the graph stages allocate and free gigabytes per sample, and fresh pages
 * cost a fault each (dense config: 0.2 s of 2.5 s per sample).  Process-wide and the host program's decision, so never done behind its
 * back: the library only does it when asked here; `build/squid`, one sample per process, does not ask.  Results are unaffected. */
int sq_keep_host_memory(void);

typedef struct sq_counts {
    int64_t n_concordant, n_blocks, n_chimeric_records, n_chim_fragments, read_len;
    int64_t n_kept_p1, n_break, n_kept_p2, n_raw_edges, n_unique_edges;
    int64_t n_order_unsolved; /* components whose ordering problem was beyond the exact solver: identity order kept, what the reference
                                 keeps when GLPK fails within its 300 s (SegmentGraph.cpp:3287-3292,3964,3984); 0 on every test input */
    int64_t token_passes_side_by_side; /* GPU reader: the largest number of token passes (one per buffer set, each on its own stream) that were
                                 running at the same time during the ingests of a STAGED file (sq_stage_bam) by this context -- 0: no such ingest (or one
                                 of fewer than eight batches); <= 4 with eight sets in flight: the HIP runtime of the process works with four hardware
                                 queues, see INTEGRATION.md (GPU_MAX_HW_QUEUES) */
    int64_t replay_candidates_checked, replay_count_mismatches; /* with SQUID_REPLAY_CHECK in the environment: break candidates of the segmentation
                                 (SegmentGraph.cpp:440-481) whose counts -- split-read support, paired-end support left / right, spanning
                                 coverage of both windows, the cluster's blocks and ConcordRest -- were recounted with the reference's linear
                                 passes, and how many of them disagreed with the counts the library had used (must be 0); else 0, 0 */
    int64_t chimeric_through_gpu_reader; /* sq_ingest_files: 1 when the chimeric BAM of the last call was inflated, cut into records and parsed on the
                                 device like the concordant one (files of 128 MiB and more; SQUID_CHIM_GPU=1 / =0 forces / forbids it), 0 when
                                 the host decoder read it.  sq_ingest_bwa_file: the same for the one file of --bwa mode, whose records need their
                                 QNAMEs too (files of 1 GiB and more; SQUID_BWA_GPU=1 / =0) */
} sq_counts;
int sq_get_counts(sq_ctx* c, sq_counts* k);

/* tests: copy the HBM-resident record SoA back into a library-owned host batch */
int sq_debug_download(sq_ctx* c, sq_aln_batch* b);
/* tests: ExactBPConcordantSupport's counting loop (src/SegmentGraph.cpp:3129-3166) over the resident records for an
 * arbitrary sorted breakpoint list; host_walk != 0 takes the serial host restatement instead of the K10 kernels */
int sq_debug_bp_support(sq_ctx* c, int32_t n_bp, const int32_t* chr, const int32_t* pos, int32_t* coverage, int32_t host_walk);
/* tests: one instance of the per-component ordering problem (GenerateILP, src/SegmentGraph.cpp:3763-3983) on local nodes
 * 0..n-1; edges5 = n_edges x {u, v, head_u, head_v, weight}, u < v.  use_gpu: k_order_small (n <= 8) / k_order_mid (9..19), else
 * the host solver (n <= 128).  Returns the canonical optimum: order[p] = local node at position p, bit-complemented (~node)
 * when the node is reversed; mask = orientation mask (bit i = node i reversed) when n <= 31, else -1. */
int sq_debug_order(sq_ctx* c, int32_t n, int32_t n_edges, const int32_t* edges5, int32_t use_gpu, int32_t* mask, int32_t* order, int64_t* value);

/* tests: the library's aligned-block comparators (SingleBamRec_t operator<, operator>, operator==, Same, CompReadPos --
 * src/SingleBamRec.h:39-58) on n blocks given as n x {RefID, RefPos, ReadPos, MatchRef, MatchRead, IsReverse, IsFirstRead}:
 * rel5 receives the five n*n relation matrices (0/1 bytes, in that order), perm_pos / perm_readpos the permutations the
 * library's sorts produce with operator< (SegmentGraph.cpp:264) and CompReadPos (ReadRec.cpp:144-145). */
int sq_debug_blocks(int32_t n, const int32_t* fields7, uint8_t* rel5, int32_t* perm_pos, int32_t* perm_readpos);
/* tuning: the two BGZF inflate kernels on the first max_blocks blocks of a file, each ALONE on the device, timed with HIP events (the
 * reader overlaps them with everything else).  variant: 2 = the lane-per-block token pass (k_inflate_tok2), else CH * 100 + PB of the
 * wave-per-block pass (k_inflate_spec: 51211, 51210, 25610, 25611, 38411, 102411).  check != 0 compares every block with zlib.
 * out7: token pass ms, resolve ms (averages over reps), inflated bytes, file bytes, blocks, tokens, blocks that differ (-1: error flag). */
int sq_debug_token_bench(sq_ctx* c, const char* path, int32_t variant, int32_t max_blocks, int32_t reps, int32_t check, double* out7);
/* tests: the RCCL transport of sq_exchange end to end on the context's device with a world of ONE rank -- librccl bound at run time,
 * ncclGetUniqueId, ncclCommInitRank(1 rank), the transport's all-gather of the fixed 16 KiB piece and of a 1 MiB remainder
 * (host -> device -> ncclAllGather -> host), bytes compared, communicator destroyed. */
int sq_debug_rccl_selftest(sq_ctx* c);

#ifdef __cplusplus
}
#endif
#endif
