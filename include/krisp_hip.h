/* krisp_hip.h -- C ABI of libkrisp_hip.so: the MI355X (gfx950) replacement for
 * krisp_fasta's k-mer generation / sort / multi-genome intersection hot path.
 *
 * The reference (grunwaldlab/krisp @ 2024_10_08) is pure Python and has no FFI
 * layer; the seams these entry points replace are (file:line into
 * /root/reference/src/krisp):
 *
 *   kr_genome_upload + kr_genome_sort   krisp_fasta/krisp_fasta.py:16-66 extractSortedKmers
 *        = kstream(...).write():        kstream/kstream.py:250-325 (generate: 617-677, 715-766,
 *                                       805-832) + sortInPlace kstream/kstream.py:83-119 (GNU sort)
 *   kr_genome_fetch_keys                the sorted "{left},{diag},{right}" k-mer file
 *                                       (krisp_fasta.py:241-243), as packed integers
 *   kr_intersect                        krisp_fasta/intersectAmplicons.py:232-310 mergeFiles
 *                                       (pairwise tree of shared.py:321-347 intersectSortedStreams)
 *                                       + filterAlignments.py:31-40 / Amplicon.py:495-521
 *   kr_cands_merge                      the same intersect applied to candidate lists that were
 *                                       produced on other GPUs (no reference counterpart: the
 *                                       reference is single-host)
 *   kr_collect + kr_fetch               the merged / filtered file: one (sequence, genome,
 *                                       multiplicity) triple per label of each line
 *                                       (Amplicon.py:170-187, 330-348)
 *
 * Conventions: plain C, no exceptions cross the boundary, every host buffer is
 * caller-allocated and caller-owned, a negative return value is an error code
 * (text via kr_last_error).  One context = one GPU = one HIP stream; a context
 * is not thread-safe, distinct contexts may be driven from distinct threads.
 * ctypes releases the GIL around each call.
 *
 * KEY FORMAT.  A window w of k = L+D+R bases (k <= 32) is split as
 * left = w[0:L], diag = w[L:L+D], right = w[L+D:k] (kstream.py:805-832 with
 * split=[L,-R]).  Its key is the string left|right|diag, 2 bits per base
 * (A=0 C=1 G=2 T=3), base j in bits 63-2j..62-2j, low bits zero.  Unsigned
 * integer order of keys == the reference's sort order (left, right, diag)
 * (GNU sort -t, -k1,1 -k3,3 with its whole-line last-resort compare).
 *
 * INPUT FORMAT.  `bases` is the ASCII text of the FASTA records of one genome,
 * records separated by ONE '\n' byte (no headers).  Windows never span a
 * separator.  Upper/lower-case ACGT, N/n are handled on the device; IUPAC
 * ambiguity letters and characters outside the reference's COMP_MAP
 * (kstream.py:11-18) must be resolved by the host before upload (the Python
 * host layer does so) -- on the device they simply invalidate their windows.
 */
#ifndef KRISP_HIP_H
#define KRISP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kr_ctx kr_ctx;

/* candidate window: (left,right) prefix present in every genome, plus, per
 * diagnostic column c (c < D <= 16) the set of bases seen at that column in
 * ingroup / outgroup genomes (bit 4c+b). */
typedef struct { uint64_t prefix, in_mask, out_mask; } kr_cand;

/* one distinct k-mer of one genome under a candidate prefix, with its multiplicity */
typedef struct { uint64_t key; uint32_t genome; uint32_t count; } kr_record;

enum {
    KR_OK = 0,
    KR_ERR_HIP = -1,       /* a HIP runtime call failed */
    KR_ERR_PARAM = -2,     /* bad argument (k > 32, D > 16, unknown genome id ...) */
    KR_ERR_CAPACITY = -3,  /* caller buffer too small / hbm budget exceeded */
    KR_ERR_STATE = -4,     /* call sequence error (e.g. intersect before sort) */
    KR_ERR_KEY = -5,       /* kr_scan_special: a surviving window holds a character outside the reference's
                              COMP_MAP -- the reference raises KeyError there (kstream.py:658) */
    KR_ERR_HOST = -6       /* kr_scan_special: bytes >= 0x80 in a candidate window, left to the host layer */
};

enum { KR_SOFT_MAP = 0,    /* lower case -> upper case (krisp_fasta default, krisp_fasta.py:33-43) */
       KR_SOFT_OMIT = 1 }; /* drop windows holding lower case (--omit-soft, krisp_fasta.py:21-31) */

/* stages with device timers (kr_stage_ms) */
enum { KR_ST_PACK = 0, KR_ST_HIST8, KR_ST_REDUCE8, KR_ST_SCATTER1, KR_ST_HIST2, KR_ST_SCAN2, KR_ST_SCATTER2,
       KR_ST_CHUNKS, KR_ST_LOCALSORT, KR_ST_FALLBACK, KR_ST_INTERSECT, KR_ST_COMPACT, KR_ST_COLLECT,
       KR_ST_MERGE, KR_ST_LOCATE, KR_ST_COUNT };   /* one kernel per stage (CHUNKS / COMPACT: two tiny ones) */

kr_ctx*     kr_create(int device, size_t hbm_budget_bytes);   /* budget 0 = no limit */
void        kr_destroy(kr_ctx*);
const char* kr_last_error(kr_ctx*);                            /* ctx may be NULL */

/* Key geometry + soft-mask rule for every genome of this context.
 * max_bases = length of the largest genome that will be uploaded (sizes the
 * radix fan-out so that all genomes share one bucket grid). */
int kr_set_params(kr_ctx*, int L, int D, int R, int softmask_mode, size_t max_bases);

/* Which strands of a window become keys (after kr_set_params, before the first upload).  BOTH is
 * kstream(complements=True), what krisp_fasta uses (krisp_fasta.py:21-43); FORWARD is neither
 * option; CANONICAL is kstream(canonicals=True): min(window, reverse complement) compared as
 * plain strings before the column split (kstream.py:679-694).  One key per window for the latter two. */
enum { KR_STRANDS_BOTH = 0, KR_STRANDS_FORWARD = 1, KR_STRANDS_CANONICAL = 2 };
int kr_set_strands(kr_ctx*, int mode);

/* kstream's --allow (kstream.py:696-713) on the device alphabet: only k-mers made of the bases whose bit
 * is set (A = 1, C = 2, G = 4, T = 8) are kept -- a generalised bad-base mask in the pack kernel.  The host
 * layer uses it when the allowed set holds no letter beyond ACGT(N) and, where both strands are emitted, is
 * closed under complement.  After kr_set_params, before the first upload. */
int kr_set_allow(kr_ctx*, unsigned base_mask);

/* kstream's --sort-cols (kstream.py:83-119: `sort -t, -kN,N ...`, then GNU sort's whole-line compare) as a key layout:
 * the window is cut into fields of widths[0..2] bases in line order (kstream.py:805-832; an empty field has width 0)
 * and the key holds them in the order order[0], order[1], order[2] (a permutation of 0 1 2), so that the unsigned
 * order of the keys is the order of that column list.  After kr_set_params(k, 0, 0, ...) with k = the sum of the
 * widths (the window as ONE field), before the first upload.  A layout moves every field by one shift, at most one
 * distinct left and one distinct right shift; the one order that needs two of a direction (2 1 0 with widths[0] !=
 * widths[2]) is served by rotating the window by a field boundary first (round 5): every permutation is taken.  Such a
 * context sorts and returns keys (kr_genome_sort, kr_genome_fetch_keys); kr_intersect and kr_collect -- whose
 * prefix is the krisp_fasta layout's -- refuse it. */
int kr_set_field_order(kr_ctx*, const int widths[3], const int order[3]);

/* ... and the column orders kr_set_field_order cannot hold (round 6): the key = the n <= 8 listed pieces of the window one after the
 * other, piece i = the widths[i] >= 1 bases at window offset offsets[i]; the pieces partition the window.  For split lists
 * with two or more sizes counted from the end (kstream.py:805-832: the line's columns then leave window order) and column
 * lists that cut the window into more than three blocks.  Same preconditions and the same sort-only context. */
int kr_set_field_pieces(kr_ctx*, int n, const int* offsets, const int* widths);

/* H2D copy of one genome's text + all device allocations it needs. */
int kr_genome_upload(kr_ctx*, int genome_id, const uint8_t* bases, size_t n_bases);
/* pack -> both-strand keys -> MSD radix partition -> LDS sort.  Asynchronous.  Under KR_OPT_LAZY_ORDER (the default) the
 * call ends behind the partition into fine buckets (<= 1600 keys on average, the bucket grid all genomes of a context
 * share); the order INSIDE the buckets -- the reference's `sort` has no such half-way state, kstream.py:83-119 -- is made
 * by the first reader that needs it: kr_genome_fetch_keys / kr_cands_probe for the whole genome, kr_intersect for its
 * anchor genome only (the other genomes' keys are looked up by hashing inside a bucket, in whatever order they lie),
 * kr_collect for the buckets its candidates touch.  Every call returns what it returns with the option off. */
int kr_genome_sort(kr_ctx*, int genome_id);
/* upload + sort + sync; returns the number of k-mer records (>= 0). */
int64_t kr_genome_add(kr_ctx*, int genome_id, const uint8_t* bases, size_t n_bases);
int64_t kr_genome_count(kr_ctx*, int genome_id);               /* syncs */
/* Adopt an already sorted key array (the content of a sorted k-mer file the reference or
 * extractSortedKmers wrote, kstream.py:832 lines packed by the host): builds the bucket grid
 * only.  Lets mergeFiles (intersectAmplicons.py:232) run on files.  Returns n. */
int64_t kr_genome_load_sorted(kr_ctx*, int genome_id, const uint64_t* keys, size_t n);
int64_t kr_genome_fetch_keys(kr_ctx*, int genome_id, uint64_t* out, size_t cap);
/* kstream WITHOUT --sort (kstream.py:327-405): the keys of an uploaded genome in stream order -- window
 * by window, a window's reverse complement right after it (KR_STRANDS_BOTH) -- no sort involved.
 * Returns the number of keys (<= 2 n_bases). */
int64_t kr_genome_keys_in_order(kr_ctx*, int genome_id, uint64_t* out, size_t cap);
int     kr_genome_free(kr_ctx*, int genome_id);

/* n-way intersection on the (left,right) prefix over sorted genomes.
 * is_ingroup[i] != 0 marks genome_ids[i] as ingroup.  apply_filter != 0 keeps
 * only candidates with a column whose ingroup and outgroup base sets are
 * disjoint (skipped when D == 0, krisp_fasta.py:265).  Returns #candidates. */
int64_t kr_intersect(kr_ctx*, const int* genome_ids, int n, const uint8_t* is_ingroup,
                     int apply_filter);
/* DNA and RNA genomes in ONE run.  The reference compares k-mers as text, and an RNA genome's are written with U
 * (kstream.py:481-508, 599): 'T' never equals 'U'.  The device alphabet has one code for both, so after this call (on != 0)
 * the diagnostic filter of every call of this context runs in its mode 2 -- a column also passes when its ingroup and
 * outgroup sets share nothing but that code (a superset of the reference's survivors, monotone like the filter itself) --
 * and the host layer, which knows each genome's alphabet, drops the (left,right) pairs that hold the code (no DNA genome
 * and RNA genome share such a pair as text) and decides the passed columns from the survivors' records
 * (krisp_amd/krisp_fasta.py: _mixed_prefix_filter, _mixed_finish).  Any time. */
int     kr_set_mixed_alphabets(kr_ctx*, int on);
int64_t kr_cands_count(kr_ctx*);
int64_t kr_cands_fetch(kr_ctx*, kr_cand* out, size_t cap);
/* Replace the candidate set (sorted by prefix, unique). */
int64_t kr_cands_load(kr_ctx*, const kr_cand* cands, size_t n);
/* candidates := candidates INTERSECT other (sorted, unique), masks OR-ed;
 * other == NULL / n == 0 with apply_filter just filters.  Returns the new count. */
int64_t kr_cands_merge(kr_ctx*, const kr_cand* other, size_t n, int have_other, int apply_filter);
/* candidates := those whose (left,right) prefix EVERY listed (sorted) genome holds, the genomes' diagnostic bases OR-ed into
 * the masks of their side (is_ingroup), the diagnostic filter applied when asked for -- what kr_intersect over these genomes
 * followed by kr_cands_merge would leave, without the genomes' own candidate list in between (for genomes of one side only
 * that list is every prefix they hold).  One step of the reference's merge tree (intersectAmplicons.py:256-307: a merged
 * file against the next genome's file) on candidate lists; the streaming flow of genome sets beyond the GPU's memory uses it
 * for every batch after the first.  Returns the count. */
int64_t kr_cands_probe(kr_ctx*, const int* genome_ids, int n, const uint8_t* is_ingroup, int apply_filter);

/* ---- multi-GPU: one process (one context) per GPU, genomes sharded over the ranks.  Sort and
 * local intersect need no communication; the ONE exchange step is a binary-tree reduction of the
 * candidate lists -- the reference's merge tree (krisp_fasta/intersectAmplicons.py:256-307: pairs of
 * sorted files merged level by level) with GPUs in place of files -- then a broadcast of the
 * survivors for the local kr_collect and a gather of the records on rank 0.  RCCL point-to-point on
 * the context's stream, device buffer to device buffer (xGMI inside a node).  Rendezvous: rank 0
 * calls kr_comm_unique_id and hands the KR_COMM_ID_BYTES bytes to the other ranks any way it likes
 * (a file, an environment variable, MPI, torch.distributed); every rank then calls kr_comm_init.
 * kr_comm_init_dir is the rehearsal transport (the same bytes through files in `dir`) for ranks
 * that share one GPU, where RCCL refuses duplicate devices: tests on a one-GPU box use it. */
#define KR_COMM_ID_BYTES 128
int     kr_comm_unique_id(uint8_t* id);
int     kr_comm_init(kr_ctx*, int rank, int world, const uint8_t* id);
int     kr_comm_init_dir(kr_ctx*, int rank, int world, const char* dir);
int     kr_comm_destroy(kr_ctx*);
int     kr_comm_rank(kr_ctx*);
int     kr_comm_world(kr_ctx*);
/* ranks of the RCCL communicator as RCCL itself counts them (ncclCommCount); 0 without one (no communicator, or
 * the file transport): bench.py prints it, so a line says whether xGMI carried the exchange */
int     kr_comm_rccl_ranks(kr_ctx*);
/* The exchange's deadline.  RCCL's calls have no timeout; every communicator of this library has a watchdog thread
 * (csrc/h_comm.inc): while an exchange call is in progress it polls ncclCommGetAsyncError, and on an asynchronous error or
 * `seconds` after the call began it aborts the communicator (ncclCommAbort) -- the call returns KR_ERR_STATE / KR_ERR_HIP,
 * later exchange calls KR_ERR_STATE; a call that still does not return ends the process with status 124.  The file
 * transport waits `seconds` for a message.  Default: the environment's KRISP_COMM_TIMEOUT, else 600.  (The reference's tree
 * runs inside one multiprocessing pool, intersectAmplicons.py:256-307: a dead worker raises there.) */
int     kr_comm_set_timeout(kr_ctx*, int seconds);
int     kr_comm_barrier(kr_ctx*);                               /* syncs the stream, then all ranks meet */
/* vals[0..n) reduced over all ranks, n <= 8: op 0 = sum, 1 = max; the result on every rank */
int     kr_comm_allreduce(kr_ctx*, double* vals, int n, int op);
/* n bytes of every rank -> every rank: out[r * n .. (r + 1) * n) = rank r's (the same n on every rank); small side
 * data of the flow (windows with IUPAC letters, krisp_fasta.py), not candidate lists */
int     kr_comm_allgather(kr_ctx*, const void* mine, size_t n, void* out);
/* candidates := those present on EVERY rank, masks OR-ed (filtered when apply_filter), on rank 0;
 * returns the count on rank 0, 0 elsewhere (the other ranks' sets are spent) */
int64_t kr_cands_reduce(kr_ctx*, int apply_filter);
int64_t kr_cands_bcast(kr_ctx*);                                /* rank 0's candidates -> every rank */
int64_t kr_records_gather(kr_ctx*);                             /* every rank's kr_collect records -> rank 0 (kr_fetch) */

/* For every candidate and every listed genome: its distinct keys + multiplicities.
 * Returns #records; kr_fetch returns them ordered by (key, position of the genome in genome_ids). */
int64_t kr_collect(kr_ctx*, const int* genome_ids, int n);
int64_t kr_fetch(kr_ctx*, kr_record* out, size_t cap);

/* ---- wide windows: amplicons longer than one 64-bit key (L+D+R > 32, or D > 16) ------------
 * The reference has no length limit (its k-mers are text: krisp_fasta.py:21-43 passes any
 * kmers= / split= to kstream, README.md:201-232 runs --conserved 30 --amplicon 100).  Here 1 <= L, R <=
 * KR_WIDE_MAX_FLANK and L+D+R <= KR_WIDE_MAX_K.  The library sorts and intersects with the 64-bit
 * pipeline the `left` spectrum, the `right` spectrum, then exact composite keys
 * rank(left):rank(right) (a flank longer than 32 bases is itself ranked through two spectra and
 * their combinations: three more sorts), applies the diagnostic filter
 * (Amplicon.py:495-521) and returns, for every surviving (left,right) group, where its member
 * windows lie; the host cuts the text of those windows out of the genomes it already holds.
 * Replaces, for such geometries, the same calls as kr_genome_sort + kr_intersect + kr_collect:
 * extractSortedKmers + mergeFiles + filterAlignments (krisp_fasta.py:237-272). */
#define KR_WIDE_MAX_K 1024
#define KR_WIDE_MAX_FLANK 256
/* one member window of a surviving group: cand = the group's number (its rank in (left,right) order
 * under KR_OPT_WIDE_ORDERED = 1; else any number, the same for all members of a group),
 * genome = index into the genome_ids of kr_wide_run, pos = base offset of the window in the
 * uploaded text, strand = 1 when the member is the reverse complement of that window */
typedef struct { uint32_t cand, genome, pos, strand; } kr_wide_hit;
int     kr_set_params_wide(kr_ctx*, int L, int D, int R, int softmask_mode, size_t max_bases);
/* genomes: kr_genome_upload.  Returns the number of hits (grouped by cand, ascending).
 * With a communicator (kr_comm_init*) the call is collective: every rank passes ITS genomes (ids unique over
 * the ranks), the flank spectra, the group list and the kept groups' mask words are exchanged inside;
 * rank 0 ends with the hits of all ranks (genome = the caller's id, not an index; no order), the others with their own. */
int64_t kr_wide_run(kr_ctx*, const int* genome_ids, int n, const uint8_t* is_ingroup, int apply_filter);
enum { KR_WIDE_DICT_LEFT = 0,   /* u64: lefts present in all genomes, sorted (L > 32: their rank:rank combinations) */
       KR_WIDE_DICT_RIGHT = 1,  /* u64: rights present in all genomes, sorted (R > 32: likewise) */
       KR_WIDE_GROUPS = 2,      /* u64: composite keys of the groups present in all genomes (before the filter) */
       KR_WIDE_HITS = 3,        /* kr_wide_hit */
       KR_WIDE_COUNTS = 4,      /* u64 per genome of the last run: its k-mer records (2 x valid windows) */
       KR_WIDE_NGROUPS = 6,     /* u64 x 1: the (left,right) groups present in all genomes (before the filter).  With L = R
                                   (and KR_OPT_WIDE_ORDERED = 0) KR_WIDE_GROUPS lists one group of every mirror pair
                                   (a:b) / (b:a) only -- the two are decided alike -- and a hit of the unlisted one
                                   carries the listed one's number with bit 31 set */
       KR_WIDE_BATCH_USED = 7,  /* u64 x 1: genomes per batch of the sort + intersect phases of the last run (0 = all at once): a genome set
                                   whose sorted keys do not fit the device goes through the phases in batches (kr_wide_run) */
       KR_WIDE_LOCATED = 8,     /* u64 x 1: member windows the locate pass of the last run LISTED (round 6: a 16-byte entry per member
                                   window instead of a group number per window start, when the members are few); 0 = the dense form ran */
       KR_WIDE_KEYS_LISTED = 9, /* u64 x 1: composite keys the last run kept as per-genome LISTS (round 6: 16 bytes per window that has a key --
                                   both flanks in the dictionaries -- instead of 16 bytes per window start, when such windows are few); 0 = dense */
       KR_WIDE_SLOT_BITS = 5 }; /* u64 x 7 (left pieces 0..2, right pieces 0..2, groups): bucket bits of the dictionary's
                                   one-sector slot table in the last run, 0 = looked up through index + sorted keys,
                                   255 = through minimizer buckets (KR_OPT_WIDE_ORDERED = 0), 254 = not built: with L = R the
                                   rights present in every genome are the reverse complements of the lefts, and a right
                                   takes the number of that left */
int64_t kr_wide_fetch(kr_ctx*, int what, void* out, size_t cap_bytes);   /* returns #elements; out == NULL: size query */
/* The member windows of the latest kr_wide_run as text, cut on the device from the genomes' uploaded bases: row i = the
 * L+D+R letters of hit i (KR_WIDE_HITS order) in line order left|diag|right, upper case, the reverse complement for a
 * strand-1 member -- what the reference writes as one line of its merged file (Amplicon.py:330-348), so the host needs no
 * copy of the genomes to render long amplicons (kr_render_windows).  One context (no communicator of more than one
 * rank: the hits of other ranks' genomes have no bases here).  Returns the number of rows; rows == NULL: size query. */
int64_t kr_wide_fetch_windows(kr_ctx*, uint8_t* rows, size_t cap_bytes);

/* Host-side ingest (no GPU involved): the text of a FASTA / sequence-per-line file -> the
 * upload buffer of kr_genome_upload, with the reference reader's semantics
 * (kstream/kstream.py:458-479 file lines, 510-537 FASTA iff the first line holds '>', 450 that line is
 * consumed when one_shot, 556-583 strip + concatenate between headers, 481-508 + 599 RNA detection and
 * U->T mapping).  universal_newlines != 0 splits on \n, \r, \r\n (plain files), else on \n
 * (gz / bz2 streams).  out needs n + 1 bytes.  stats[4] = records, characters outside
 * ACGTNacgtn (the host resolves those), is_rna (-1 undecided, 0, 1), is_fasta.  Returns the
 * number of bytes written. */
int64_t kr_fasta_to_bases(const uint8_t* text, size_t n, int universal_newlines, int one_shot, uint8_t* out,
                          size_t cap, int64_t* stats);

/* Host-side ingest of one FILE: read -> inflate (.gz: libdeflate when the box
 * has it, else zlib; every member of a multi-member file, BGZF members side by side on host threads,
 * one large member cut into chunks that decode side by side -- csrc/h_pgzip.inc; .bz2: libbz2,
 * block by block on host threads) -> kr_fasta_to_bases, with the reference
 * reader's semantics for files (kstream.py:458-479: .gz by extension; 510-583).  *bases = a buffer
 * the library owns -- pinned host memory when a GPU is present, so that kr_genome_upload copies
 * from it by DMA -- until kr_host_free.  stats[8] = records, characters outside ACGTNacgtn, is_rna,
 * is_fasta, read us, inflate us, parse us, gzip members | used_libdeflate << 32.  Returns the
 * number of bytes; a .bz2 file on a box without libbz2 returns KR_ERR_HOST (the host layer inflates it). */
int64_t kr_ingest_file(const char* path, uint8_t** bases, int64_t* stats);
/* The same ingest with the PARSE on the device (the host reads and inflates only):
 *   kr_read_file            the file's text in pinned host memory (free with kr_host_free); stats[8]: [3] = 1 when the
 *                           text wants universal newlines (a plain file), [4] read us, [5] inflate us, [6] copy us,
 *                           [7] gzip members | used_libdeflate << 32.  Returns the number of bytes (.bz2 without libbz2: KR_ERR_HOST).
 *   kr_genome_upload_text   text -> genome `id`'s upload buffer, parsed on the device (csrc/k_text.inc) with exactly
 *                           kr_fasta_to_bases' result (the tests compare them byte for byte), then as
 *                           kr_genome_upload.  stats[4] = records, characters outside ACGTNacgtn, is_rna, is_fasta.
 *                           Returns the number of bases (separators included).
 *   kr_genome_fetch_bases   an uploaded genome's bases back to the host (the IUPAC side channel, text cutting) */
int64_t kr_read_file(const char* path, uint8_t** text, int64_t* stats);
/* Round 6 -- a BGZF file (bgzip: a chain of gzip members of <= 64 KB of text that say
 * how long they are) inflated ON THE DEVICE, a lane per member (csrc/k_inflate.inc), then parsed there as
 * kr_genome_upload_text parses a text (.gz streams: no universal newlines): `file` = the bytes of the file as they lie on
 * disk.  Every member's CRC-32 and ISIZE are checked on the device.  Returns the number of bases; KR_ERR_HOST -- nothing
 * uploaded -- when the file is not BGZF all the way (plain gzip, other extra fields, >= 2^32 bytes) or a member does not
 * inflate to its trailer: the caller then reads the file as any other `.gz` (kr_read_file), and that path's verdict on it
 * stands (the reference: kstream.py:458-479, gzip.open).  stats[8]: [0..3] as kr_genome_upload_text, [4] members,
 * [5] first bad member, [6] its status, [7] microseconds of the inflate + CRC kernels. */
int64_t kr_genome_upload_bgzf(kr_ctx*, int id, const uint8_t* file, size_t n, int one_shot, int64_t* stats);
/* The large device buffers of `n` genomes (ids[]) of up to n_bases bases each -- upload buffer, sorted-key arrays, sort
 * lanes; with_text: the device reader's copy of the file text as well -- made ahead of the uploads, while host threads
 * still read and inflate the files (krisp_fasta.py:86-123 starts a process per genome; here the one context gets its
 * memory while the files inflate: memory another process has just given back costs 15-40 ms per GB to obtain, seconds at 3 Gbp).  After kr_set_params;
 * purely an optimisation: the uploads and sorts make or grow whatever is missing. */
int kr_reserve(kr_ctx*, const int* ids, int n, size_t n_bases, int with_text);
/* out4[0] = bytes this context may still allocate (the device's free memory, or the rest of its HBM budget when that is
 * less), [1] = the device's total, [2] = bytes the context holds, [3] = its budget (0 = none).  The command line sizes its
 * genome batches with it when a genome set does not fit one GPU (krisp_amd/krisp_fasta.py: the streaming flow -- the
 * reference has no such limit: kstream.py:108-119 sorts in external memory, intersectAmplicons.py:232-310 merges files) */
int kr_mem_info(kr_ctx*, int64_t* out4);
int64_t kr_genome_upload_text(kr_ctx*, int id, const uint8_t* text, size_t n, int universal_newlines, int one_shot,
                              int64_t* stats);
int64_t kr_genome_fetch_bases(kr_ctx*, int id, uint8_t* out, size_t cap);
void    kr_host_free(void* p);

/* Host-side (no GPU involved): the side channel of SURVEY 8(b) -- the windows the device's 2-bit
 * alphabet cannot carry.  `bases` is an upload buffer (kr_fasta_to_bases).  Returns the number of
 * window starts (ascending) of windows of k characters that survive the soft-mask rule, hold no
 * N/n and hold an IUPAC ambiguity letter: the reference keeps those k-mers and complements them
 * letter by letter (COMP_MAP kstream.py:11-18, _complements 644-677); the binder cuts them out
 * of its text and joins them with the device results.  KR_ERR_KEY (+ *bad_char) where the
 * reference raises KeyError (kstream.py:658).  starts may be NULL (count only). */
int64_t kr_scan_special(const uint8_t* bases, size_t n, int k, int omit_soft, uint64_t* starts, size_t cap,
                        int* bad_char);

/* Final text from survivor records of packed keys -- replaces Amplicon.py:523-671 (render_alignment, render_csv,
 * makeBracket, consensus) and outputAlignments.py:26-162 at --cores 1 for large results: the CSV and the alignment
 * blocks as bytes, one pass, no object per Amplicon (krisp_amd/amplicon.py keeps the general path; the tests run
 * both and compare byte for byte).  recs ordered by (key, label id); label_of[genome] = rank of the genome's label
 * among the distinct labels in string order; label_in[label] = 1 for ingroup labels, NULL when no outgroup was
 * given.  Host only.  Returns the number of groups, KR_ERR_HOST for a group this code leaves to the general path
 * (no all-ingroup row for the CSV consensus: the reference raises there).  Free the texts with kr_text_free. */
int64_t kr_render_records(const kr_record* recs, size_t n, int L, int D, int R, const uint32_t* label_of,
                          size_t n_genomes, const char* const* label_text, size_t n_labels, const uint8_t* label_in,
                          int dot, char** csv, size_t* csv_len, char** align, size_t* align_len);
/* ... and from member windows of long amplicons (kr_wide_fetch_windows; Amplicon.py:598-671 on text the reference holds as
 * strings anyway): rows = n windows of L+D+R letters in line order, cand[i] = group number of row i (the rows of a group
 * adjacent: KR_WIDE_HITS order), genome[i] = index into label_of.  Groups are ordered by (left, right), members by (diag,
 * label) here.  rna: write T as U.  KR_ERR_HOST also when one (left,right) arrives under two group numbers. */
int64_t kr_render_windows(const uint8_t* rows, size_t n, int L, int D, int R, const uint32_t* cand, const uint32_t* genome,
                          const uint32_t* label_of, size_t n_genomes, const char* const* label_text, size_t n_labels,
                          const uint8_t* label_in, int dot, int rna, char** csv, size_t* csv_len, char** align,
                          size_t* align_len);
void    kr_text_free(void* p);

/* Options that change HOW (never what) the library computes; results are identical for every
 * value (tests run the suite under each).  Set after kr_create, before kr_set_params. */
enum { KR_OPT_SLICE_BASES = 1,       /* -1 automatic; 0..4: sort every genome in 4^n key-space slices (the
                                        large-genome path, pass 0 + per-slice pass 1) */
       KR_OPT_GENERIC_INTERSECT = 2, /* 1: every intersect sub-tile takes the generic path (key-count
                                        splits, ranges from the prefixes) instead of whole-bucket sub-tiles */
       KR_OPT_ISECT_FORMAT = 3,      /* 0 automatic; 1: the narrow per-prefix state also for D <= 4; 2: automatic, but never the
                                        32-bit word per prefix that one diagnostic column and <= 24 genomes otherwise get */
       KR_OPT_ABLATE = 4,            /* timing aids of kr_debug_*; refused unless built with -DKR_ABLATE */
       KR_OPT_WIDE_SLOTS = 5,        /* wide path: 1 (default) dictionaries also as one-sector slot tables, 0 index + sorted keys only */
       KR_OPT_PLACE_TRIES = 7,       /* 1 .. 16 (default 8): candidate allocations of the pass-1 output buffer (>= 256 MB), each timed under
                                        pass 1's write pattern, the fastest kept: physical placement moves pass 1 / pass 2 by up to 15 %
                                        from one allocation to the next; worth its ~5 ms per candidate for contexts that sort many genomes */
       KR_OPT_ISECT_KERNEL = 8,      /* 0 (default) the persistent, pipelined intersect kernels over items of whole buckets wherever a
                                        (left,right) group lies inside one fine bucket (with 32-bit heads where the geometry allows);
                                        1: one workgroup per chunk everywhere; 2: pipelined with 64-bit heads only */
       KR_OPT_LANES = 9,             /* sort lanes, 1 .. 8: consecutive kr_genome_sort calls go to consecutive lanes (own stream, own
                                        scratch: 16 bytes per base each) and overlap on the device (3 lanes: -7 % per step on 4 x 50
                                        Mbp); kr_intersect and every call that reads a sorted genome join them.  0 (default):
                                        automatic -- one lane until the context has sorted 16 genomes (a lane costs ~10 ms to set
                                        up, more than a one-shot run gets back), 3 from then on.  Key-space slices use one lane
                                        whatever the setting, kr_wide_run two when more than one is asked for.  May be set at any time */
       KR_OPT_LAZY_ORDER = 10,       /* 1 (default): kr_genome_sort stops at the fine buckets, the LDS sort runs where a reader needs
                                        the order (see kr_genome_sort); 0: every sort ends with it (rounds 1-5) */
       KR_OPT_WIDE_ORDERED = 6 };    /* wide path: 0 (default) flanks of >= 20 bases are numbered through minimizer buckets (look-ups
                                        of neighbouring windows share memory sectors): the same groups and hits, but `cand` no longer
                                        ascends with (left, right); 1: order-preserving ranks, groups in the reference's order */
int     kr_set_option(kr_ctx*, int option, int64_t value);

int     kr_sync(kr_ctx*);
/* HIP-event timers on the context's stream */
int     kr_timer_begin(kr_ctx*);
double  kr_timer_end_ms(kr_ctx*);                              /* syncs */
/* accumulated device time of one stage since the last kr_stage_reset (syncs);
 * only collected after kr_stage_enable(ctx, 1) (it serialises launches). */
int     kr_stage_enable(kr_ctx*, int on);
/* time only the stages whose bit (1 << KR_ST_x) is set: the event pairs around every launch
 * cost ~4 % of a C2 step when all stages are timed; bench.py times the dominant kernel only
 * inside its timed region (the full table comes from a calibration pass before it) */
int     kr_stage_select(kr_ctx*, unsigned stage_mask);
int     kr_stage_reset(kr_ctx*);
double  kr_stage_ms(kr_ctx*, int stage);
int64_t kr_stage_launches(kr_ctx*, int stage);

/* introspection for stage-level parity tests: 0 codes(u64) 1 bad(u32) 2 hist(u32)
 * 3 bucket offsets(u32) 4 keys after pass 1 5 keys after pass 2.  Returns the
 * number of ELEMENTS copied (of the element type above). */
int64_t kr_debug_fetch(kr_ctx*, int genome_id, int what, void* out, size_t cap_bytes);
/* property check for sizes no oracle reaches: adjacent key pairs out of order (0 = sorted) */
int64_t kr_debug_inversions(kr_ctx*, int genome_id);
/* timing aid: the LDS sort re-run `reps` times over a sorted genome (mode must be 0).  Average ms per launch. */
double  kr_debug_localsort(kr_ctx*, int genome_id, int reps, int mode);
/* timing aid: the intersect kernel `reps` times (mode 0 as shipped; a -DKR_ABLATE build also
 * takes 64 keys streamed but not probed, 128 probed without the LDS update, ...: k_intersect.inc);
 * leaves the candidate set invalid */
double  kr_debug_intersect(kr_ctx*, const int* genome_ids, int n, const uint8_t* is_ingroup, int reps, int mode);
/* measured streaming-copy rate of this device (read + write GB/s): bench.py reports the roofline
 * fraction against it beside the 8 TB/s specification figure */
double  kr_debug_copy_gbps(kr_ctx*, size_t bytes, int reps);
/* which copy form and grid gave that figure (1 / 2 / 4 loads in flight per lane, plain or non-temporal, 2 .. 32 workgroups
 * per CU are all timed; the best is reported): a string owned by the context, valid until the next probe */
const char* kr_debug_copy_which(kr_ctx*);
/* the pipelined intersect kernels: items that went to the chunk kernel (oversized), slices redone by chunks,
 * threads per workgroup, log2(buckets per item) and 1 = 32-bit heads of the latest launch; out[5] = sort lanes in use
 * now; out[6..7] = intersections that left their late genomes to the probe, candidates the latest of them probed */
/* 0 for a product build; else the result-changing experiment switches the library was compiled with (bit 0 -DKR_EXPERIMENTS,
 * 1 KR_ABLATE, 2 I3_ABL, 3 P2_ABL, 4 LS2_NORANK, 5 KR_EXP_NOLOOKUP, 6 KR_EXP_MZONLY: csrc/k_keys.inc).  No context, no GPU needed. */
int     kr_build_experiments(void);
/* KR_OPT_LAZY_ORDER's counters: out[0] LDS sorts of whole slices kr_genome_sort left out, [1] made later (anchor, fetch, probe),
 * [2] kr_collect calls that read only the buckets their candidates touch, [3] the option's value, [4] intersection launches whose
 * anchor genome stayed in bucket order as well (k_intersect3t<., UA>), [5] 1 unless KR_FUSE_ANCHOR=0, [6..7] 0 */
int     kr_debug_lazy(kr_ctx*, int64_t* out8);
int     kr_debug_isect(kr_ctx*, int64_t* out8);
/* the latest placement search of the pass-1 output buffers: out8[0] candidates probed, [1] buffers handed to the sort lanes,
 * [2..5] probe milliseconds of the four fastest candidates, [6] the median, [7] the slowest (bench.py prints them: which
 * placement class a line ran in) */
int     kr_debug_place(kr_ctx*, double* out8);
/* what the multi-GPU exchange has cost this context so far: out[0] host synchronisations, [1] point-to-point calls,
 * [2] collectives, [3] microseconds inside kr_cands_reduce / kr_cands_bcast, [4] kr_cands_reduce calls, [5] entries of the
 * agreed message size */
int     kr_debug_comm(kr_ctx*, int64_t* out8);
/* microseconds per blocking call of the exchange on this box (host clock, RCCL communicator of any world size, every rank
 * calls it): out3[0] all-reduce of three doubles + synchronisation, [1] a send / receive pair of `bytes` to the rank itself
 * + synchronisation, [2] a broadcast of `bytes` + synchronisation */
int     kr_debug_comm_probe(kr_ctx*, size_t bytes, int reps, double* out3);
/* one round of the tree with the rank itself as its partner over RCCL: the candidate list + header sent to and received from
 * itself, merged as a received list is (count read on the device).  What a one-GPU box can show of the exchange on the real
 * transport; returns the count (unchanged for an unfiltered list) */
int64_t kr_debug_cands_selfexchange(kr_ctx*, int apply_filter);
/* test aid for the communicator's watchdog: a receive nobody sends to + the synchronisation behind it; returns when the
 * watchdog has aborted the communicator (KR_ERR_STATE) or RCCL has refused the call (KR_ERR_HIP); the communicator is spent */
int     kr_debug_comm_hang(kr_ctx*);
/* test aids: bytes left of the context's HBM budget (-1 = no budget); make `left` bytes remain from now on */
int64_t kr_debug_budget_left(kr_ctx*);
int     kr_debug_budget_set(kr_ctx*, int64_t left);
int     kr_debug_info(kr_ctx*, int64_t* out8);  /* b, nbuckets, T, CAP, nwg, overflow segments, fallback launches, 0 */

#ifdef __cplusplus
}
#endif
#endif
