// The device-resident merge engine: .fast_mnn / .fast_mnn_core (R/fastMNN.R:398-562) with the merge tree of
// R/MNN_tree.R, driven from one host thread.  Host code only decides sizes and control flow (the merge order, k,
// whether a merge is skipped); every cell x dimension value stays in HBM from upload to download.
#pragma once
#include <functional>
#include <memory>

#include "bmx_common.hpp"
#include "bmx_ops.hpp"

namespace bmx {

struct Segment {
    int batch;  // 1-based original batch id
    int n;      // rows
};

struct Node {
    std::vector<int> index;  // batch ids in this node, in row order (MNN_treenode@index)
    DevBuf<double> data;     // [n][d] row-major
    int n = 0;
    bool has_restrict = false;
    DevBuf<int32_t> restrict_rows;  // 0-based, in the caller's order (any R subsetting vector: any order, cells may repeat)
    int n_restrict = 0;
    // a cell named more than once is that many POINTS to the searches and to the centring mean (R/MNN_tree.R:113-127,
    // R/fastMNN.R:633-637) but ONE cell to .average_correction's rowsum (R/fastMNN.R:571-579): positions of the same cell
    // are chained -- dup_next[r] = the next position of the cell at position r (-1: none), dup_head[r] = 1 at its first
    bool restrict_dups = false;
    DevBuf<int32_t> dup_next, dup_head;
    std::vector<Segment> origin;  // MNN_treenode@origin as run lengths
    std::vector<int> stat_slot;   // per segment: slot of its current column means / total variance, -1 = stale
    std::vector<int> extras;      // ids of batch vectors in the engine's pool (MNN_treenode@extras)
};

struct TreeSlot {
    int left = -1, right = -1;  // children (indices into the slot vector), -1 for a leaf
    std::unique_ptr<Node> node; // set for finished nodes (materialised leaves or finished merges)
    int batch = -1;             // a leaf that has not been materialised yet: its batch (0-based) ...
    int64_t arena_row = 0;      // ... and its first row in the run's arena
    bool ready() const { return (bool)node || batch >= 0; }
};

struct MergeRecord {
    std::vector<int> left_set, right_set;
    DevBuf<int32_t> first, second;  // 1-based rows within the left / right node at merge time
    int64_t npairs = 0;
    int64_t stats[6] = {0, 0, 0, 0, 0, 0};
    double batch_size = 0.0;
    bool batch_size_na = true;
    bool skipped = false;
    int64_t asv_tally[3] = {-1, -1, -1};  // testing hook "asv_modes": cells this merge's tiled adjust_shift_variance re-ran / flagged beyond the re-run / handled
    int bs_slot = -1;              // slot of batch.size in the device scalar buffer
    std::vector<int> var_batches;  // per segment (left segments first): batch id, ...
    std::vector<int> old_slot, new_slot;  // ... slots of its total variance before / after the merge's centring
};

void bmx_shard_range_impl(int64_t n, int rank, int world, int64_t* begin, int64_t* end);
int64_t bmx_shard_rows_per_rank(int64_t n, int world);

class Engine {
  public:
    explicit Engine(int device);
    ~Engine();
    void set_shard(int rank, int world, bmx_allgather_fn fn, void* ctx);
    // production exchange: an RCCL communicator owned by the engine, all-gathers in place on the engine's stream
    void init_rccl(int rank, int world, const void* unique_id);
    // exchange statistics since the last run() started
    void emulate(int mode, int rank, int world);  // 0 off, 1 record (single rank), 2 replay as rank of world (see engine.hip)
    int64_t exchange_calls() const { return xchg_calls_; }
    int64_t exchange_bytes() const { return xchg_bytes_; }
    // lazy: the matrices are NOT copied here -- the caller keeps them valid until the next run() has returned, which
    // pulls each batch through the pinned staging ring on a copy stream when its leaf comes up and prefetches the next
    // ones while the GPU is busy with a search (the one-shot bmx_fast_mnn: upload hidden behind the first merges)
    void upload(int nbatches, int d, const double* const* data, const int32_t* nrows,
                const int32_t* const* restrict_idx, const int32_t* n_restrict, bool lazy = false);
    void run(const bmx_params_t& p, const int32_t* tree, int tree_len);
    struct OptimisticRetry {};  // thrown by the waits of an optimistic run whose device flag is up (see run)
    void download(double* corrected, int32_t* batch, int32_t* merge_left, int32_t* merge_right, double* batch_size,
                  int32_t* skipped, double* lost_var);
    void pairs(int merge, int32_t** left, int32_t** right, int64_t* npairs);
    int64_t pairs_count(int merge) const;
    void pairs_into(int merge, int32_t* left, int32_t* right);  // caller-allocated, pairs_count(merge) entries each
    // every merge's lists in one pass of the host threads (capacity[m] >= pairs_count(m))
    void pairs_all_into(int nmerges, int32_t* const* left, int32_t* const* right, const int64_t* capacity);
    void merge_stats(int merge, int64_t* out6) const;
    void set_profiling(bool on) { knn_ws_.profile = on; }
    // Host-side watchdog (bmx_common.hpp: guarded_stream_sync): every wait of a run has a deadline of base_s plus a term
    // scaled from the work queued; base_s <= 0 switches it off.  Default 60 s (BMX_WATCHDOG_MS overrides).
    void set_watchdog(double base_s) { wd_base_s_ = base_s; }
    bool dead() const { return dead_; }
    // testing hook: a kernel that keeps the engine's stream busy for `ms` milliseconds and then ends by itself
    void debug_stall(int ms);
    // diagnostics: keep a copy of the two matrices merge `merge` searches (left and right node after
    // orthogonalisation, R/fastMNN.R:473-477); -1 = off
    void set_snapshot(int merge) { snap_merge_ = merge; }
    void snapshot(double* left_rm, double* right_rm, int64_t* nl, int64_t* nr);
    void profile_var_adj(double* out3);  // {ms, launches, (cell, restricted cell) pairs} of the run's adjust_shift_variance calls
    void snapshot_var_adj(double* left_rm, double* right_rm, double* corr_rm, double* scaling, int32_t* r1, int32_t* r2,
                          int64_t* sizes4);
    // testing hook "asv_modes" on during the run: per merge the tiled form's tallies (re-run, flagged beyond it, handled; -1:
    // not recorded), and the way every right cell of the snapshot merge went (dst[0, n): 0 / 1 / 2, 255 beyond the record)
    void var_adj_tally(int merge, int64_t* out3) const;
    void snapshot_var_adj_modes(unsigned char* dst, int64_t n) const;
    void profile(double* topk_ms, int64_t* launches, int64_t* fallbacks);
    // out[0..9]: full-pass ms / launches of the fp16 kernel, of the split-bf16 kernel, sample-pass ms / launches,
    // streaming-section ms (everything of the merges that is not a kNN search), queries that took the exact path,
    // queries handed from tier 1 to tier 2, all since the last run() started (profiling on); out[9]: runs of this engine
    // whose optimistic attempt gave up and was repeated with host-checked searches
    void profile_detail(double* out10);
    int nbatches() const { return B_; }
    int64_t total_cells() const { return N_; }
    hipStream_t stream() const { return stream_; }

    // single primitives (also used by the host-pointer parity entry points)
    void knn(const double* X, const int32_t* ref_rows, int nr, const double* Q, const int32_t* q_rows, int nq, int k,
             int32_t* idx, double* dist, const float* seed_d2 = nullptr, const double* centre = nullptr,
             double* kth = nullptr, bool gather = true);
    struct MnnOut {
        int64_t P = 0;
        int U = 0;
        int k1 = 0, k2 = 0;
        int nsel = 0;  // left cells that occur in some right cell's list = rows of idxLR_ (see lsel_)
    };
    // findMutualNN on (restricted) left / right rows; leaves idxLR_ (one row per SELECTED left cell, lsel_), idxRL_,
    // cntL_, offL_ (per row of idxLR_), partR_, cntR_, second_u_
    MnnOut find_mnn(const Node& left, const Node& right, int k, double prop_k, const double* mu_left = nullptr,
                    const double* mu_right = nullptr);

  private:
    DevBlockCache cache_;  // first member: destroyed last, after every DevBuf below has handed its block back
  public:
    DevBlockCache* cache() { return &cache_; }
    int d_ = 0;
    KnnWorkspace knn_ws_;
    ScanWorkspace scan_ws_;
    ReduceWorkspace red_ws_;
    DevBuf<int32_t> idxLR_, idxRL_, cntL_, offL_, partR_, cntR_, offR_, second_u_, second_rows_, idxT_;
    DevBuf<int32_t> stampL_, offSel_, lsel_, qsel_, cntFold_;
    int64_t optimistic_retries_ = 0;  // runs of this engine that started over with host-checked searches (run())
    int state_seq_ = 0;  // sequence number of the last publish_state (read_state)
    int stamp_gen_ = 0;  // number of the last search whose listed rows were stamped (stampL_ is never cleared)
    DevBuf<unsigned long long> maskL_;
    SortedRows sorted_;  // k2 > 64: both neighbour lists, each row sorted (pairs.hip)
    DevBuf<double> kthL_;  // per selected left cell: the largest distance of its row of idxLR_ (+inf where unknown)
    DevBuf<double> distT_, distRL_, averaged_, loc_, vecs_, scal_, means_pool_;
    DevBuf<float> seedL_;
    DevBuf<double> corr_, asv_ws_, asv_scale_;
    DevBuf<double> arena_;  // [N][d]: the rows of a predefined-tree run, leaves in tree order (root_ aliases it)
    DevBuf<int32_t> iota_l_, iota_r_;
    int n_slots_ = 0, slot_cap_ = 0;

  private:
    void run_once(const bmx_params_t& p, const int32_t* tree, int tree_len);
    // auto-merge: n independent pair counts dealt round robin over the ranks (each an unsharded search), then all-gathered
    std::vector<int32_t> solo_counts(int n, const std::function<int(int)>& count, bool dry = false);
    std::vector<int32_t> gather_counts(const std::vector<int32_t>& mine);
    DevBuf<int32_t> count_xchg_;
    DevBuf<int32_t> xflags_;   // [world][4] the ranks' optimistic-search flags, gathered with every search's lists
    bool solo_ = false;        // this rank is running a whole search of its own (auto-merge counts): read_state does not restart
    bool solo_flag_ = false;   // ... but remembers that it should have; gather_counts tells every rank
    const int32_t* read_state();  // one wait: the run's device words in pinned memory; throws OptimisticRetry
    void merge_step(int mdx, Node& left, Node& right, const bmx_params_t& p, std::unique_ptr<Node>& merged);
    // statistics (column means + total variance) of the segments whose slot is stale
    void ensure_stats(Node& node);
    void ensure_stats2(Node& a, Node* b);  // both nodes of a merge in one pass
    void node_means(const Node& left, const Node& right, double* mu_l, double* mu_r);  // one launch where both are fresh
    void centre_both(Node& left, Node& right, int vid, const double* mu_l, const double* mu_r);
    // one pass over the node: centre along the given batch vectors (may be none), optionally with fresh statistics
    // mu_known: the node's restrict-row column mean when the caller already has it (it is invariant under centring)
    void row_pass(Node& node, const std::vector<int>& vec_ids, bool with_stats, const double* mu_known = nullptr);
    void node_mean(const Node& node, double* mu);  // column mean over the restrict rows (all rows without restrict)
    void orthogonalize(Node& node, const std::vector<int>& extras);
    int count_mnn_pairs(const Node& left, const Node& right, const bmx_params_t& p);
    std::unique_ptr<Node> clone_node(const Node& src);
    void exchange(void* buf, int64_t bytes_per_rank, bool flags_only = false);

    void wait(double work_s = 0.0);  // guarded wait on the engine's stream: deadline wd_base_s_ + work_s
    void ensure_uploaded(int b);   // batch b's copy is queued (lazy uploads: on the copy stream, with an event)
    void prefetch_one();           // queue the next batch the merge order will need, if any is still on the host
    std::vector<const double*> host_data_;  // lazy upload: the caller's matrices (valid until run() returns)
    std::vector<char> uploaded_;
    std::vector<hipEvent_t> up_ev_;
    hipStream_t copy_stream_ = nullptr;
    std::vector<int> need_order_;
    size_t need_pos_ = 0;
    bool lazy_ = false;
    // Pair lists of the last run in OUTPUT-row numbering, all merges in one block (merge m: left ids at pairs_off_[m],
    // right ids behind them), copied to pinned host memory at the end of the run: pairs_into is then a host copy
    void stage_pairs();
    DevBuf<int32_t> pairs_all_, pairs_tab_;
    std::vector<int64_t> pairs_off_;
    std::vector<int32_t> pairs_tab_host_;
    void* pairs_pin_ = nullptr;
    size_t pairs_pin_bytes_ = 0;
    bool pairs_pinned_ = false;  // the last run's lists are (on their way) in pairs_pin_
    hipEvent_t pairs_ev_ = nullptr, pairs_ready_ev_ = nullptr;
    bool pairs_copy_pending_ = false;
    void check_alive() const;
    void mark_dead();
    double wd_base_s_ = 60.0;
    double queued_work_s_ = 0.0;  // watchdog allowance of what was queued since the last wait that came back
    double* scal_pin_ = nullptr;  // pinned landing area of the end-of-run scalar read-back
    size_t scal_pin_bytes_ = 0;
    bool dead_ = false;

    int device_ = 0;
    hipStream_t stream_ = nullptr;
    int rank_ = 0, world_ = 1;
    bmx_allgather_fn gather_fn_ = nullptr;
    void* gather_ctx_ = nullptr;
    void* comm_ = nullptr;  // ncclComm_t
    int64_t xchg_calls_ = 0, xchg_bytes_ = 0;
    int emu_mode_ = 0;                                        // bmx_engine_emulate
    std::vector<std::pair<DevBuf<char>, int64_t>> emu_rec_;   // what each exchange of the recorded run would have gathered
    size_t emu_next_ = 0;
    int64_t fallbacks_ = 0;
    struct Section {  // a streaming section bracketed by events while profiling
        Engine* e;
        std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
        explicit Section(Engine* eng);
        ~Section();
    };
    int snap_merge_ = -1;
    DevBuf<double> snap_l_, snap_r_;
    int64_t snap_nl_ = 0, snap_nr_ = 0;
    DevBuf<double> snap_al_, snap_ar_, snap_ac_, snap_as_;  // the snapshot merge's variance adjustment: inputs and scalings
    DevBuf<int32_t> snap_ai1_, snap_ai2_;
    int64_t snap_anl_ = 0, snap_anr_ = 0, snap_ar1_ = 0, snap_ar2_ = 0;
    std::vector<unsigned char> snap_modes_;
    double asv_pairs_ = 0.0;  // (cell, restricted cell) pairs of this rank's adjust_shift_variance calls in the last run

    int B_ = 0;
    int64_t N_ = 0;
    std::vector<int> nrows_;
    std::vector<DevBuf<double>> inputs_cm_;        // resident inputs, column-major as uploaded
    std::vector<DevBuf<int32_t>> inputs_restrict_; // 0-based
    std::vector<DevBuf<int32_t>> inputs_dup_;      // [2 n_restrict]: dup_next, dup_head of a batch whose restrict repeats cells
    std::vector<int> n_restrict_;                  // -1 = NULL
    std::vector<char> has_dups_;

    // results of the last run
    std::unique_ptr<Node> root_;
    std::vector<MergeRecord> merges_;
    int n_extras_ = 0;
    std::vector<double> scal_host_;
    friend struct EngineAccess;
};

}  // namespace bmx
