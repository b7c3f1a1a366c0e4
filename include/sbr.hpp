// sbr.hpp — C++17 host side of the MI355X sequence-recommender engine.
//
// The reference crate (maciejkula/sbr-rs) is Rust; no Rust toolchain exists in this image, so the
// host layer above the C-ABI (include/sbr_hip.h, libsbr_hip.so) is written in C++ and mirrors the
// crate's public interface for the sequence-model path: same module layout (sbr::data,
// sbr::models::lstm, sbr::models::ewma, sbr::evaluation, sbr::datasets), same names, same
// argument meaning, same error behaviour, so that the reference's own tests read the same here
// (tests/cpp/facade_tests.cpp).  Header-only; needs nothing but sbr_hip.h and libsbr_hip.so.
//
//   reference                                  here
//   Result<T, E>                               sbr::Result<T, E>  (is_ok / is_err / unwrap / unwrap_err)
//   panic!                                     sbr::EngineError (std::runtime_error)
//   rand::XorShiftRng                          sbr::XorShiftRng   (the engine's documented stream)
//   Vec<f32>, &[ItemId]                        std::vector<float>, const std::vector<ItemId>&
//
// Everything numeric runs on the GPU through the C-ABI.  There is no CPU fallback: without a
// gfx950 device `build()` throws EngineError(SBR_ERR_NO_DEVICE).
#ifndef SBR_HPP
#define SBR_HPP

#include <algorithm>
#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <limits>
#include <memory>
#include <numeric>
#include <optional>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <variant>
#include <vector>

#include "sbr_hip.h"

namespace sbr {

// ---- lib.rs:77-116 ---------------------------------------------------------------------------
using UserId = std::size_t;    // lib.rs:77
using ItemId = std::size_t;    // lib.rs:79
using Timestamp = std::size_t; // lib.rs:81

/// Prediction error types (lib.rs:84-89).
enum class PredictionError { InvalidPredictionValue };
/// Fitting error types (lib.rs:93-97).
enum class FittingError { NoInteractions };

inline const char* to_string(PredictionError) { return "Invalid prediction value: non-finite or not a number."; }
inline const char* to_string(FittingError) { return "No interactions were supplied."; }

/// Anything the reference would `panic!` on, plus ABI-level failures (no device, HIP error, out
/// of memory, unsupported configuration).
struct EngineError : std::runtime_error {
    sbr_status status;
    EngineError(sbr_status st, const std::string& where)
        : std::runtime_error(where + ": " + sbr_status_string(st)), status(st) {}
};

/// The engine keeps the scratch of a fit call (device and pinned-host blocks up to 64 MiB) for the next call — the reference's
/// own bench re-fits one model in a loop (benches/benchmark.rs:40-42).  A long-lived process that is done fitting gives it back.
inline void release_cached_memory() { sbr_release_cached_memory(); }

/// Minimal Result: holds either a value or an error enum.
template <class T, class E>
class Result {
  public:
    static Result Ok(T value) { return Result(std::in_place_index<0>, std::move(value)); }
    static Result Err(E error) { return Result(std::in_place_index<1>, error); }
    bool is_ok() const { return v_.index() == 0; }
    bool is_err() const { return v_.index() == 1; }
    /// The value; throws std::runtime_error with the error text otherwise (≙ `unwrap` panicking).
    T& unwrap() {
        if (is_err()) throw std::runtime_error(std::string("called unwrap() on an Err value: ") + to_string(std::get<1>(v_)));
        return std::get<0>(v_);
    }
    const T& unwrap() const { return const_cast<Result*>(this)->unwrap(); }
    E unwrap_err() const {
        if (is_ok()) throw std::runtime_error("called unwrap_err() on an Ok value");
        return std::get<1>(v_);
    }

  private:
    template <std::size_t I, class U>
    Result(std::in_place_index_t<I> tag, U&& u) : v_(tag, std::forward<U>(u)) {}
    std::variant<T, E> v_;
};

// ---- the index RNG --------------------------------------------------------------------------
/// rand 0.5's `XorShiftRng` as recalled: Marsaglia xorshift128, next_u64 = low word
/// first, gen_range / shuffle / Uniform with the crate's widening-multiply rejection zones — the
/// object handed to `Hyperparameters::rng` / `from_seed` (lstm.rs:122-132).  Identical in the library
/// (sbr_numerics.h: sbr_xorshift, sbr_rand_*), the Python host layer and here; the crate's source is not
/// in this image and no reference test pins a stream.
class XorShiftRng {
  public:
    static XorShiftRng from_seed(const std::array<std::uint8_t, 16>& seed) {
        XorShiftRng r;
        bool any = false;
        for (int i = 0; i < 4; ++i) {
            r.s_[i] = (std::uint32_t)seed[4 * i] | ((std::uint32_t)seed[4 * i + 1] << 8) |
                      ((std::uint32_t)seed[4 * i + 2] << 16) | ((std::uint32_t)seed[4 * i + 3] << 24);
            any = any || r.s_[i] != 0;
        }
        if (!any) r.s_ = {0x193A6754u, 0xA8A7D469u, 0x97830E05u, 0x113BA7BBu};
        return r;
    }
    /// `XorShiftRng::from_seed(rand::thread_rng().gen())` (lstm.rs:67)
    static XorShiftRng from_entropy() {
        std::random_device rd;
        std::array<std::uint8_t, 16> seed;
        for (auto& b : seed) b = (std::uint8_t)rd();
        return from_seed(seed);
    }
    /// The 16 bytes that re-create the current state through from_seed.
    std::array<std::uint8_t, 16> state_seed() const {
        std::array<std::uint8_t, 16> out;
        for (int i = 0; i < 4; ++i)
            for (int b = 0; b < 4; ++b) out[4 * i + b] = (std::uint8_t)(s_[i] >> (8 * b));
        return out;
    }
    std::uint32_t next_u32() {
        const std::uint32_t t = s_[0] ^ (s_[0] << 11);
        s_[0] = s_[1];
        s_[1] = s_[2];
        s_[2] = s_[3];
        s_[3] = s_[3] ^ (s_[3] >> 19) ^ (t ^ (t >> 8));
        return s_[3];
    }
    std::uint64_t next_u64() {
        const std::uint64_t lo = next_u32();
        const std::uint64_t hi = next_u32();
        return lo | (hi << 32);
    }
    /// `Rng::gen_range(low, high)` (rand 0.5 `UniformInt::sample_single`): zone = range << leading_zeros(range),
    /// accept when the low half of next_u64() * range is <= zone, result = low + high half.
    /// Panics in the crate when low >= high (`assert!(low < high)`); here: std::invalid_argument.
    std::uint64_t gen_range(std::uint64_t low, std::uint64_t high) {
        if (low >= high) throw std::invalid_argument("gen_range: low must be below high");
        const std::uint64_t range = high - low;
        const std::uint64_t zone = range << __builtin_clzll(range);
        for (;;) {
            const unsigned __int128 m = (unsigned __int128)next_u64() * range;
            if ((std::uint64_t)m <= zone) return low + (std::uint64_t)(m >> 64);
        }
    }
    /// `Uniform::new(low, high).sample(rng)`: zone = MAX - (MAX - range + 1) % range (data.rs:77-78).
    std::uint64_t uniform(std::uint64_t low, std::uint64_t high) {
        if (low >= high) throw std::invalid_argument("Uniform::new: low must be below high");
        const std::uint64_t range = high - low, max = std::numeric_limits<std::uint64_t>::max();
        const std::uint64_t zone = max - (max - range + 1) % range;
        for (;;) {
            const unsigned __int128 m = (unsigned __int128)next_u64() * range;
            if ((std::uint64_t)m <= zone) return low + (std::uint64_t)(m >> 64);
        }
    }
    /// gen_range(0, n)
    std::uint64_t below(std::uint64_t n) { return gen_range(0, n); }
    /// `rng.gen::<f64>()` (rand 0.5 `Standard`): 53 random bits scaled to [0, 1).
    double unit() { return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }
    /// `Rng::shuffle`: i = len; while i >= 2 { i -= 1; swap(i, gen_range(0, i + 1)) }.
    template <class T>
    void shuffle(std::vector<T>& v) {
        for (std::size_t i = v.size(); i >= 2;) {
            i -= 1;
            std::swap(v[i], v[gen_range(0, i + 1)]);
        }
    }

  private:
    std::array<std::uint32_t, 4> s_{};
};

// =============================================================================================
// sbr::data  (src/data.rs)
// =============================================================================================
namespace data {

/// Basic interaction type (data.rs:17-51).
class Interaction {
  public:
    Interaction(UserId user_id, ItemId item_id, Timestamp timestamp)
        : user_id_(user_id), item_id_(item_id), timestamp_(timestamp) {}
    UserId user_id() const { return user_id_; }
    ItemId item_id() const { return item_id_; }
    float weight() const { return 1.0f; }
    Timestamp timestamp() const { return timestamp_; }
    bool operator==(const Interaction& o) const {
        return user_id_ == o.user_id_ && item_id_ == o.item_id_ && timestamp_ == o.timestamp_;
    }

  private:
    UserId user_id_;
    ItemId item_id_;
    Timestamp timestamp_;
};

class CompressedInteractions;
class TripletInteractions;

/// A collection of individual interactions (data.rs:92-211).
class Interactions {
  public:
    Interactions(std::size_t num_users, std::size_t num_items) : num_users_(num_users), num_items_(num_items) {}
    /// `impl From<Vec<Interaction>>` (data.rs:200-211): num_users / num_items = max id + 1.
    static Interactions from(std::vector<Interaction> interactions) {
        std::size_t nu = 0, ni = 0;
        for (const auto& x : interactions) {
            nu = std::max(nu, x.user_id() + 1);
            ni = std::max(ni, x.item_id() + 1);
        }
        Interactions out(nu, ni);
        out.interactions_ = std::move(interactions);
        return out;
    }
    void push(const Interaction& interaction) { interactions_.push_back(interaction); }
    const std::vector<Interaction>& data() const { return interactions_; }
    std::size_t len() const { return interactions_.size(); }
    bool is_empty() const { return interactions_.empty(); }
    void shuffle(XorShiftRng& rng) { rng.shuffle(interactions_); }
    std::pair<Interactions, Interactions> split_at(std::size_t idx) const {
        Interactions head(num_users_, num_items_), tail(num_users_, num_items_);
        head.interactions_.assign(interactions_.begin(), interactions_.begin() + (std::ptrdiff_t)idx);
        tail.interactions_.assign(interactions_.begin() + (std::ptrdiff_t)idx, interactions_.end());
        return {std::move(head), std::move(tail)};
    }
    /// (those for which func is true, the rest) — data.rs:149-172
    template <class F>
    std::pair<Interactions, Interactions> split_by(F func) const {
        Interactions head(num_users_, num_items_), tail(num_users_, num_items_);
        for (const auto& x : interactions_) (func(x) ? head : tail).interactions_.push_back(x);
        return {std::move(head), std::move(tail)};
    }
    inline CompressedInteractions to_compressed() const;
    inline TripletInteractions to_triplet() const;
    std::size_t num_users() const { return num_users_; }
    std::size_t num_items() const { return num_items_; }
    std::pair<std::size_t, std::size_t> shape() const { return {num_users_, num_items_}; }

  private:
    std::size_t num_users_, num_items_;
    std::vector<Interaction> interactions_;
};

/// Random split (data.rs:54-64): shuffles `interactions` in place; the first test_fraction of the
/// shuffled data is the test set.  Returns (train, test).
inline std::pair<Interactions, Interactions> train_test_split(Interactions& interactions, XorShiftRng& rng,
                                                              float test_fraction) {
    interactions.shuffle(rng);
    const std::size_t cut = (std::size_t)(test_fraction * (float)interactions.len());
    auto parts = interactions.split_at(cut);
    return {std::move(parts.second), std::move(parts.first)};
}

namespace detail {
/// SipHash-2-4 of one u64 written as 8 little-endian bytes (`Hasher::write_usize` on a 64-bit
/// target, data.rs:81-84; siphasher 0.2's SipHasher is SipHash-2-4).
inline std::uint64_t siphash24_u64(std::uint64_t k0, std::uint64_t k1, std::uint64_t m) {
    auto rotl = [](std::uint64_t x, int b) { return (x << b) | (x >> (64 - b)); };
    std::uint64_t v0 = k0 ^ 0x736F6D6570736575ull, v1 = k1 ^ 0x646F72616E646F6Dull;
    std::uint64_t v2 = k0 ^ 0x6C7967656E657261ull, v3 = k1 ^ 0x7465646279746573ull;
    auto round = [&]() {
        v0 += v1; v1 = rotl(v1, 13); v1 ^= v0; v0 = rotl(v0, 32);
        v2 += v3; v3 = rotl(v3, 16); v3 ^= v2;
        v0 += v3; v3 = rotl(v3, 21); v3 ^= v0;
        v2 += v1; v1 = rotl(v1, 17); v1 ^= v2; v2 = rotl(v2, 32);
    };
    v3 ^= m; round(); round(); v0 ^= m;
    const std::uint64_t tail = (std::uint64_t)8 << 56; // message length in the top byte, no tail bytes
    v3 ^= tail; round(); round(); v0 ^= tail;
    v2 ^= 0xFF;
    round(); round(); round(); round();
    return v0 ^ v1 ^ v2 ^ v3;
}
} // namespace detail

/// Split so that no user is in both sets (data.rs:69-88): two u64 keys from rng, SipHash-2-4 of
/// the user id, train iff hash % 100000 > (test_fraction * 100000) as u64.  Returns (train, test).
inline std::pair<Interactions, Interactions> user_based_split(Interactions& interactions, XorShiftRng& rng,
                                                              float test_fraction) {
    const std::uint64_t denominator = 100000;
    const std::uint64_t train_cutoff = (std::uint64_t)(test_fraction * (float)denominator);
    const std::uint64_t key_0 = rng.uniform(0, std::numeric_limits<std::uint64_t>::max());
    const std::uint64_t key_1 = rng.uniform(0, std::numeric_limits<std::uint64_t>::max());
    return interactions.split_by([&](const Interaction& x) {
        return detail::siphash24_u64(key_0, key_1, (std::uint64_t)x.user_id()) % denominator > train_cutoff;
    });
}

/// A single user's data, earliest to latest (data.rs:339-371).  A view into CompressedInteractions.
struct CompressedInteractionsUser {
    UserId user_id;
    const std::uint32_t* item_ids;
    const std::uint64_t* timestamps;
    std::size_t length;

    std::size_t len() const { return length; }
    bool is_empty() const { return length == 0; }
    /// One chunk of at most chunk_size interactions: (item_ids, timestamps).
    using Chunk = std::pair<std::vector<ItemId>, std::vector<Timestamp>>;
    /// Chunked view: the FIRST chunk is the short one, all the others have chunk_size elements
    /// (data.rs:363-370, 406-431).
    std::vector<Chunk> chunks(std::size_t chunk_size) const {
        std::vector<Chunk> out;
        std::size_t idx = 0;
        while (idx < length) {
            const std::size_t rem = (length - idx) % chunk_size;
            const std::size_t size = rem == 0 ? chunk_size : rem;
            out.emplace_back(std::vector<ItemId>(item_ids + idx, item_ids + idx + size),
                             std::vector<Timestamp>(timestamps + idx, timestamps + idx + size));
            idx += size;
        }
        return out;
    }
};

/// CSR by user, time-sorted (data.rs:227-329).  Storage is exactly what the C-ABI takes:
/// user_pointers u64 [num_users + 1], item_ids u32 [nnz] (sbr_model_fit / sbr_mrr_score).
class CompressedInteractions {
  public:
    CompressedInteractions(std::size_t num_users, std::size_t num_items, std::vector<std::uint64_t> user_pointers,
                           std::vector<std::uint32_t> item_ids, std::vector<std::uint64_t> timestamps)
        : num_users_(num_users), num_items_(num_items), user_pointers_(std::move(user_pointers)),
          item_ids_(std::move(item_ids)), timestamps_(std::move(timestamps)) {}

    /// `impl From<&Interactions>` (data.rs:236-265): stable sort by (user, timestamp).
    static CompressedInteractions from(const Interactions& interactions) {
        std::vector<Interaction> data = interactions.data();
        std::stable_sort(data.begin(), data.end(), [](const Interaction& a, const Interaction& b) {
            return a.user_id() != b.user_id() ? a.user_id() < b.user_id() : a.timestamp() < b.timestamp();
        });
        std::vector<std::uint64_t> ptr(interactions.num_users() + 1, 0);
        std::vector<std::uint32_t> items;
        std::vector<std::uint64_t> ts;
        items.reserve(data.size());
        ts.reserve(data.size());
        for (const auto& x : data) {
            ptr[x.user_id() + 1] += 1;
            items.push_back((std::uint32_t)x.item_id());
            ts.push_back((std::uint64_t)x.timestamp());
        }
        for (std::size_t u = 0; u < interactions.num_users(); ++u) ptr[u + 1] += ptr[u];
        return CompressedInteractions(interactions.num_users(), interactions.num_items(), std::move(ptr), std::move(items),
                                      std::move(ts));
    }

    std::optional<CompressedInteractionsUser> get_user(UserId user_id) const {
        if (user_id >= num_users_) return std::nullopt;
        const std::size_t start = (std::size_t)user_pointers_[user_id], stop = (std::size_t)user_pointers_[user_id + 1];
        return CompressedInteractionsUser{user_id, item_ids_.data() + start, timestamps_.data() + start, stop - start};
    }
    /// Iterate over users (data.rs:269-274).
    std::vector<CompressedInteractionsUser> iter_users() const {
        std::vector<CompressedInteractionsUser> out;
        out.reserve(num_users_);
        for (UserId u = 0; u < num_users_; ++u) out.push_back(*get_user(u));
        return out;
    }
    std::size_t num_users() const { return num_users_; }
    std::size_t num_items() const { return num_items_; }
    std::pair<std::size_t, std::size_t> shape() const { return {num_users_, num_items_}; }
    Interactions to_interactions() const {
        Interactions out(num_users_, num_items_);
        for (UserId u = 0; u < num_users_; ++u)
            for (std::uint64_t k = user_pointers_[u]; k < user_pointers_[u + 1]; ++k)
                out.push(Interaction(u, item_ids_[k], (Timestamp)timestamps_[k]));
        return out;
    }
    const std::vector<std::uint64_t>& user_pointers() const { return user_pointers_; }
    const std::vector<std::uint32_t>& item_ids() const { return item_ids_; }
    const std::vector<std::uint64_t>& timestamps() const { return timestamps_; }

  private:
    std::size_t num_users_, num_items_;
    std::vector<std::uint64_t> user_pointers_;
    std::vector<std::uint32_t> item_ids_;
    std::vector<std::uint64_t> timestamps_;
};

inline CompressedInteractions Interactions::to_compressed() const { return CompressedInteractions::from(*this); }

/// A minibatch of triplet interactions: views into the three arrays (data.rs:502-520).
struct TripletMinibatch {
    const UserId* user_ids;
    const ItemId* item_ids;
    const Timestamp* timestamps;
    std::size_t size;
    std::size_t len() const { return size; }
    bool is_empty() const { return size == 0; }
};

/// Interactions in COO form (data.rs:435-481).  Not consumed by the sequence models; kept for API parity.
class TripletInteractions {
  public:
    /// Minibatches of exactly minibatch_size interactions over [idx, stop_idx): a shorter remainder is never
    /// yielded (data.rs:522-545).  next() returns false at the end.
    class MinibatchIterator {
      public:
        MinibatchIterator(const TripletInteractions* interactions, std::size_t idx, std::size_t stop_idx, std::size_t minibatch_size)
            : interactions_(interactions), idx_(idx), stop_idx_(stop_idx), minibatch_size_(minibatch_size) {}
        /// The same data and minibatch size over [start, stop) (data.rs:491-499).
        MinibatchIterator slice(std::size_t start, std::size_t stop) const { return {interactions_, start, stop, minibatch_size_}; }
        bool next(TripletMinibatch& out) {
            const std::size_t start = idx_, stop = idx_ + minibatch_size_;
            idx_ = stop;
            if (stop > stop_idx_) return false;
            out = {interactions_->user_ids_.data() + start, interactions_->item_ids_.data() + start,
                   interactions_->timestamps_.data() + start, minibatch_size_};
            return true;
        }

      private:
        const TripletInteractions* interactions_;
        std::size_t idx_, stop_idx_, minibatch_size_;
    };

    /// `impl From<&Interactions>` (data.rs:558-575): the interactions' own order.
    static TripletInteractions from(const Interactions& interactions) {
        TripletInteractions out;
        out.num_users_ = interactions.num_users();
        out.num_items_ = interactions.num_items();
        out.user_ids_.reserve(interactions.len());
        out.item_ids_.reserve(interactions.len());
        out.timestamps_.reserve(interactions.len());
        for (const auto& x : interactions.data()) {
            out.user_ids_.push_back(x.user_id());
            out.item_ids_.push_back(x.item_id());
            out.timestamps_.push_back(x.timestamp());
        }
        return out;
    }
    std::size_t len() const { return user_ids_.size(); }
    bool is_empty() const { return user_ids_.empty(); }
    MinibatchIterator iter_minibatch(std::size_t minibatch_size) const { return {this, 0, len(), minibatch_size}; }
    /// num_partitions iterators over consecutive slices of len / num_partitions interactions (integer division:
    /// the remainder belongs to no partition, data.rs:463-475).
    std::vector<MinibatchIterator> iter_minibatch_partitioned(std::size_t minibatch_size, std::size_t num_partitions) const {
        const MinibatchIterator iterator = iter_minibatch(minibatch_size);
        const std::size_t chunk_size = len() / num_partitions;
        std::vector<MinibatchIterator> out;
        for (std::size_t x = 0; x < num_partitions; ++x) out.push_back(iterator.slice(x * chunk_size, (x + 1) * chunk_size));
        return out;
    }
    std::size_t num_users() const { return num_users_; }
    std::size_t num_items() const { return num_items_; }
    std::pair<std::size_t, std::size_t> shape() const { return {num_users_, num_items_}; }

  private:
    std::size_t num_users_ = 0, num_items_ = 0;
    std::vector<UserId> user_ids_;
    std::vector<ItemId> item_ids_;
    std::vector<Timestamp> timestamps_;
};

inline TripletInteractions Interactions::to_triplet() const { return TripletInteractions::from(*this); }

} // namespace data

// =============================================================================================
// OnlineRankingModel (lib.rs:101-116)
// =============================================================================================
template <class UserRepresentation>
class OnlineRankingModel {
  public:
    virtual ~OnlineRankingModel() = default;
    /// From a chronologically ordered sequence of items compute a user representation.
    virtual Result<UserRepresentation, PredictionError> user_representation(const std::vector<ItemId>& item_ids) const = 0;
    /// Scores of `item_ids` for that user representation.
    virtual Result<std::vector<float>, PredictionError> predict(const UserRepresentation& user,
                                                                const std::vector<ItemId>& item_ids) const = 0;
};

// =============================================================================================
// sbr::models  (src/models/mod.rs, sequence_model.rs, lstm.rs, ewma.rs)
// =============================================================================================
namespace models {

/// The user representation used by implicit sequence models (mod.rs:9-12).
struct ImplicitUser {
    std::vector<float> user_embedding;
};
/// The loss used for training the model (mod.rs:15-23).
enum class Loss { BPR = SBR_LOSS_BPR, Hinge = SBR_LOSS_HINGE, WARP = SBR_LOSS_WARP };
/// Optimizer used to train the model (mod.rs:26-32).
enum class Optimizer { Adagrad = SBR_OPT_ADAGRAD, Adam = SBR_OPT_ADAM };
/// Type of parallelism (mod.rs:35-41).  Synchronous: every replica sees every update before its
/// next minibatch.  Asynchronous: the deterministic analogue of Hogwild — with more than one replica,
/// minibatch k+1 is computed on parameters that lack update k (staleness exactly one step), so the
/// exchange runs underneath the computation.
enum class Parallelism { Asynchronous = SBR_PAR_ASYNCHRONOUS, Synchronous = SBR_PAR_SYNCHRONOUS };

namespace detail {

inline void check(sbr_status st, const char* where) {
    if (st != SBR_OK) throw EngineError(st, where);
}

inline std::vector<std::uint32_t> narrow(const std::vector<ItemId>& ids) {
    std::vector<std::uint32_t> out(ids.size());
    for (std::size_t i = 0; i < ids.size(); ++i) out[i] = (std::uint32_t)ids[i];
    return out;
}

/// Owner of the device replicas behind one model object: `num_threads` replicas (≙ the worker
/// threads of sequence_model.rs:90-102), replica r on HIP device r mod device_count
/// (sbr_group_create).
class Replicas {
  public:
    Replicas(sbr_hparams hp, bool partition_item_table) : hp_(hp), partitioned_(partition_item_table) {
        handles_.assign(hp.num_devices, nullptr);
        const sbr_status st = sbr_group_create(&hp, hp.num_devices, partition_item_table ? SBR_GROUP_PARTITION_ITEM_TABLE : 0u,
                                               handles_.data());
        if (st != SBR_OK) {
            handles_.clear();
            throw EngineError(st, "sbr_group_create");
        }
    }
    Replicas(const Replicas&) = delete;
    Replicas& operator=(const Replicas&) = delete;
    ~Replicas() { release(); }

    sbr_model* primary() const { return handles_[0]; }
    const sbr_hparams& hparams() const { return hp_; }
    const std::vector<sbr_model*>& handles() const { return handles_; }
    bool partitioned() const { return partitioned_; }

    Result<float, FittingError> fit(const data::CompressedInteractions& interactions) {
        float loss = 0.0f;
        const sbr_status st =
            sbr_group_fit(handles_.data(), (std::uint32_t)handles_.size(), interactions.user_pointers().data(),
                          interactions.item_ids().data(), (std::uint64_t)interactions.num_users(), &loss);
        if (st == SBR_ERR_NO_INTERACTIONS) return Result<float, FittingError>::Err(FittingError::NoInteractions);
        check(st, "sbr_group_fit");
        return Result<float, FittingError>::Ok(loss);
    }

  private:
    void release() {
        for (sbr_model* h : handles_)
            if (h) sbr_model_destroy(h);
        handles_.clear();
    }
    sbr_hparams hp_;
    bool partitioned_ = false;
    std::vector<sbr_model*> handles_;
};

/// fit + OnlineRankingModel over the C-ABI, shared by both model types
/// (lstm.rs:391-416, ewma.rs:404-429 → sequence_model.rs:70-232).
class ImplicitSequenceModel : public OnlineRankingModel<ImplicitUser> {
  public:
    ImplicitSequenceModel(const sbr_hparams& hp, bool partition_item_table)
        : replicas_(std::make_unique<Replicas>(hp, partition_item_table)) {}

    /// Fit the model; re-callable (training continues).  Err(NoInteractions) when no subsequence
    /// of more than two items exists (sequence_model.rs:86-88).
    Result<float, FittingError> fit(const data::CompressedInteractions& interactions) { return replicas_->fit(interactions); }

    /// The crate's own ORDER of work at one subsequence per step (`batch_sequences(1)`, embedding_dim <= 32): a step's negatives
    /// from the worker's sequential XorShiftRng stream (sequence_model.rs:58-65, :137) and, with `num_threads(n)`, one optimiser
    /// application per worker in worker order (:163-166) — instead of the engine's counter-keyed draws and summed gradients.
    /// Call before `fit`.  Throws EngineError(SBR_ERR_UNSUPPORTED) outside the mode's shapes.
    void set_reference_order(bool on) {
        for (sbr_model* h : replicas_->handles()) {
            const sbr_status st = sbr_model_set_reference_order(h, on ? 1 : 0);
            if (st != SBR_OK) throw EngineError(st, "sbr_model_set_reference_order");
        }
    }

    /// The number the reference's `fit` would have returned for the last `fit` call: sequence_model.rs:157 reads the
    /// loss node BEFORE :160 runs its forward pass; the nodes are shared running sums (lstm.rs:322-328), so a subsequence
    /// of s steps contributes L_{s-1} of the worker's most recent earlier subsequence with at least s steps.  `fit` itself returns the true mean loss.
    float last_fit_lagged_loss() const {
        float v = 0.0f;
        check(sbr_model_last_fit_lagged_loss(replicas_->primary(), &v), "sbr_model_last_fit_lagged_loss");
        return v;
    }

    Result<ImplicitUser, PredictionError> user_representation(const std::vector<ItemId>& item_ids) const override {
        const std::vector<std::uint32_t> ids = narrow(item_ids);
        ImplicitUser user{std::vector<float>(replicas_->hparams().embedding_dim)};
        const sbr_status st = sbr_user_representation(replicas_->primary(), ids.data(), ids.size(), user.user_embedding.data());
        if (st == SBR_ERR_INVALID_PREDICTION)
            return Result<ImplicitUser, PredictionError>::Err(PredictionError::InvalidPredictionValue);
        check(st, "sbr_user_representation");
        return Result<ImplicitUser, PredictionError>::Ok(std::move(user));
    }

    Result<std::vector<float>, PredictionError> predict(const ImplicitUser& user,
                                                        const std::vector<ItemId>& item_ids) const override {
        if (user.user_embedding.size() != replicas_->hparams().embedding_dim)
            throw EngineError(SBR_ERR_INVALID_ARGUMENT, "predict: user embedding has the wrong dimension");
        const std::vector<std::uint32_t> ids = narrow(item_ids);
        std::vector<float> out(ids.size());
        const sbr_status st = sbr_predict(replicas_->primary(), user.user_embedding.data(), ids.data(), ids.size(), out.data());
        if (st == SBR_ERR_INVALID_PREDICTION)
            return Result<std::vector<float>, PredictionError>::Err(PredictionError::InvalidPredictionValue);
        check(st, "sbr_predict");
        return Result<std::vector<float>, PredictionError>::Ok(std::move(out));
    }

    /// The engine handle (replica 0), for evaluation's fused path and for parameter access.
    sbr_model* handle() const { return replicas_->primary(); }
    const sbr_hparams& hparams() const { return replicas_->hparams(); }

    /// One parameter block (sbr_param) — what the serde derives at lstm.rs:204,386 expose.
    std::vector<float> parameter(sbr_param which) const {
        std::uint64_t count = 0;
        check(sbr_model_param_count(handle(), which, &count), "sbr_model_param_count");
        std::vector<float> out(count);
        if (count) check(sbr_model_get_param(handle(), which, out.data(), count), "sbr_model_get_param");
        return out;
    }

    /// ≙ the serde derives on Hyperparameters / Parameters / the models (lstm.rs:38,204,386; ewma.rs:44,208,401; `bincode`
    /// in Cargo.toml:18): hyper-parameters, every parameter and optimiser-state block, the two counters and the state of
    /// the model RNG — everything the next `fit` depends on — in one little-endian file:
    ///   "SBRM" u32 version  u32 sizeof(sbr_hparams)  sbr_hparams  u8 partitioned  u64 epoch  u64 optimiser steps
    ///   u8 rng[16]  u32 blocks  { i32 which  u64 count  f32[count] } ...
    /// A model built with num_threads(n) stores ONE copy (replicas are bit-identical by construction).
    void save(const std::string& path) const {
        std::ofstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("save: cannot open " + path);
        auto put = [&](const void* p, std::size_t n) { f.write(reinterpret_cast<const char*>(p), (std::streamsize)n); };
        const std::uint32_t version = 1, hp_bytes = (std::uint32_t)sizeof(sbr_hparams);
        put("SBRM", 4); put(&version, 4); put(&hp_bytes, 4);
        const sbr_hparams hp = replicas_->hparams();
        put(&hp, sizeof hp);
        const std::uint8_t part = replicas_->partitioned() ? 1 : 0;
        put(&part, 1);
        std::uint64_t epoch = 0, steps = 0;
        check(sbr_model_get_counters(handle(), &epoch, &steps), "sbr_model_get_counters");
        put(&epoch, 8); put(&steps, 8);
        std::uint8_t rng[16];
        check(sbr_model_get_rng(handle(), rng), "sbr_model_get_rng");
        put(rng, 16);
        std::vector<std::pair<std::int32_t, std::vector<float>>> blocks;
        for (int w = SBR_PARAM_ITEM_EMBEDDING; w <= SBR_PARAM_EWMA_ALPHA_M; ++w) {
            std::vector<float> v = parameter((sbr_param)w);
            if (!v.empty()) blocks.emplace_back((std::int32_t)w, std::move(v));
        }
        const std::uint32_t nb = (std::uint32_t)blocks.size();
        put(&nb, 4);
        for (const auto& b : blocks) {
            const std::uint64_t count = b.second.size();
            put(&b.first, 4); put(&count, 8); put(b.second.data(), count * sizeof(float));
        }
        if (!f) throw std::runtime_error("save: write failed: " + path);
    }

  protected:
    /// Rebuilds the replicas of a saved model (one copy of a partitioned table, never n full tables) and restores the
    /// saved state into every one of them: the next `fit` is the one the saved model would have run.
    explicit ImplicitSequenceModel(const std::string& path, int expect_ewma) {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("load: cannot open " + path);
        auto get = [&](void* p, std::size_t n) {
            f.read(reinterpret_cast<char*>(p), (std::streamsize)n);
            if (!f) throw std::runtime_error("load: truncated file: " + path);
        };
        char magic[4];
        std::uint32_t version = 0, hp_bytes = 0;
        get(magic, 4); get(&version, 4); get(&hp_bytes, 4);
        if (std::string(magic, 4) != "SBRM" || version != 1 || hp_bytes != sizeof(sbr_hparams))
            throw std::runtime_error("load: not a model file of this library version: " + path);
        sbr_hparams hp;
        get(&hp, sizeof hp);
        if ((hp.model == SBR_MODEL_EWMA) != (expect_ewma != 0)) throw std::runtime_error("load: the file holds the other model type: " + path);
        // a corrupt header must not size allocations or build replicas (sbr_model_create re-checks the rest)
        if (hp.num_items == 0 || hp.embedding_dim == 0 || hp.embedding_dim > 256 || hp.num_devices == 0 || hp.num_devices > 16 ||
            hp.device_rank != 0 || hp.batch_sequences == 0 || hp.max_sequence_length < 3 || hp.model < 0 || hp.model > 2)
            throw std::runtime_error("load: implausible hyper-parameters in " + path);
        std::uint8_t part = 0;
        get(&part, 1);
        std::uint64_t epoch = 0, steps = 0;
        get(&epoch, 8); get(&steps, 8);
        std::uint8_t rng[16];
        get(rng, 16);
        replicas_ = std::make_unique<Replicas>(hp, part != 0);
        std::uint32_t nb = 0;
        get(&nb, 4);
        if (nb > SBR_PARAM_EWMA_ALPHA_M + 1) throw std::runtime_error("load: implausible block count in " + path);
        std::uint32_t seen = 0;
        for (std::uint32_t i = 0; i < nb; ++i) {
            std::int32_t which = 0;
            std::uint64_t count = 0, expect = 0;
            get(&which, 4); get(&count, 8);
            // the engine says how many elements block `which` of THIS model has: a corrupt count never sizes an allocation
            if (which < 0 || which > SBR_PARAM_EWMA_ALPHA_M || (seen >> which & 1u) ||
                sbr_model_param_count(replicas_->primary(), which, &expect) != SBR_OK || expect == 0 || count != expect)
                throw std::runtime_error("load: parameter block " + std::to_string(which) + " does not fit the model in " + path);
            seen |= 1u << which;
            std::vector<float> v(count);
            get(v.data(), count * sizeof(float));
            const bool table = which == SBR_PARAM_ITEM_EMBEDDING || which == SBR_PARAM_ITEM_EMBEDDING_ACC || which == SBR_PARAM_ITEM_BIAS ||
                               which == SBR_PARAM_ITEM_BIAS_ACC || which == SBR_PARAM_ITEM_EMBEDDING_M || which == SBR_PARAM_ITEM_BIAS_M;
            for (std::size_t r = 0; r < replicas_->handles().size(); ++r) {
                if (part && table && r > 0) continue;  // a partitioned table exists once: written through replica 0
                check(sbr_model_set_param(replicas_->handles()[r], (sbr_param)which, v.data(), count), "sbr_model_set_param");
            }
        }
        for (std::int32_t which = 0; which <= SBR_PARAM_EWMA_ALPHA_M; ++which) {  // every non-empty block of the model must have been in the file
            std::uint64_t expect = 0;
            if (sbr_model_param_count(replicas_->primary(), which, &expect) == SBR_OK && expect != 0 && !(seen >> which & 1u))
                throw std::runtime_error("load: parameter block " + std::to_string(which) + " is missing from " + path);
        }
        for (sbr_model* h : replicas_->handles()) {
            check(sbr_model_set_counters(h, epoch, steps), "sbr_model_set_counters");
            check(sbr_model_set_rng(h, rng), "sbr_model_set_rng");
        }
    }

  private:
    std::unique_ptr<Replicas> replicas_;
};

/// Builder state common to lstm::Hyperparameters (lstm.rs:39-202) and ewma::Hyperparameters
/// (ewma.rs:45-206); defaults are those of `new` (lstm.rs:56-71, ewma.rs:61-75).
template <class Derived>
class HyperparametersBase {
  public:
    HyperparametersBase(std::size_t num_items, std::size_t max_sequence_length)
        : num_items_(num_items), max_sequence_length_(max_sequence_length), rng_(XorShiftRng::from_entropy()) {}

    /// Set the learning rate.
    Derived& learning_rate(float v) { learning_rate_ = v; return self(); }
    /// Set the L2 penalty.
    Derived& l2_penalty(float v) { l2_penalty_ = v; return self(); }
    /// Set the embedding dimensionality (the engine supports 16, 32, 64, 128, 256).
    Derived& embedding_dim(std::size_t v) { item_embedding_dim_ = v; return self(); }
    /// Set the number of epochs to run per each `fit` call.
    Derived& num_epochs(std::size_t v) { num_epochs_ = v; return self(); }
    /// Set the loss function.
    Derived& loss(Loss v) { loss_ = v; return self(); }
    /// Number of parallel workers = device replicas (≙ rayon threads, lstm.rs:110-113).  The
    /// reference default is one per core; here the default is one replica.
    Derived& num_threads(std::size_t v) { num_threads_ = v; return self(); }
    /// Set the type of parallelism.
    Derived& parallelism(Parallelism v) { parallelism_ = v; return self(); }
    /// Set the random number generator (moved in, as in the reference).
    Derived& rng(XorShiftRng v) { rng_ = v; return self(); }
    /// Set the random number generator from seed.
    Derived& from_seed(const std::array<std::uint8_t, 16>& seed) { rng_ = XorShiftRng::from_seed(seed); return self(); }
    /// Set the optimizer type.
    Derived& optimizer(Optimizer v) { optimizer_ = v; return self(); }
    /// Subsequences per optimiser step and device (engine extension; 1 = the reference's
    /// per-sequence SGD).
    Derived& batch_sequences(std::size_t v) { batch_sequences_ = v; return self(); }
    /// With num_threads(n) > 1: store the item table once, row range r on replica r's device, instead
    /// of n full copies (engine extension for catalogues in the 1e7 range; results are identical).
    Derived& partition_item_table(bool v) { partition_item_table_ = v; return self(); }

  protected:
    Derived& self() { return static_cast<Derived&>(*this); }

    sbr_hparams hparams(sbr_model_kind kind) const {
        sbr_hparams hp{};
        hp.num_items = (std::uint32_t)num_items_;
        hp.max_sequence_length = (std::uint32_t)max_sequence_length_;
        hp.embedding_dim = (std::uint32_t)item_embedding_dim_;
        hp.learning_rate = learning_rate_;
        hp.l2_penalty = l2_penalty_;
        hp.model = kind;
        hp.loss = (std::int32_t)loss_;
        hp.optimizer = (std::int32_t)optimizer_;
        hp.parallelism = (std::int32_t)parallelism_;
        const auto seed = rng_.state_seed();
        std::copy(seed.begin(), seed.end(), hp.seed);
        hp.num_epochs = (std::uint32_t)num_epochs_;
        hp.num_devices = (std::uint32_t)num_threads_;
        hp.device_rank = 0;
        hp.batch_sequences = (std::uint32_t)batch_sequences_;
        return hp;
    }

    /// The parts of `random` the two models share (lstm.rs:141-172, ewma.rs:141-165): same ranges
    /// as the reference; the draws come from this engine's RNG.
    void randomize_common(XorShiftRng& rng) {
        auto uniform = [&](double lo, double hi) { return lo + (hi - lo) * rng.unit(); };
        max_sequence_length_ = (std::size_t)1 << (4 + rng.below(4));
        item_embedding_dim_ = (std::size_t)1 << (4 + rng.below(4));
        learning_rate_ = std::pow(10.0f, (float)uniform(-3.0, 0.5));
        l2_penalty_ = std::pow(10.0f, (float)uniform(-7.0, -3.0));
        loss_ = uniform(0.0, 1.0) < 0.5 ? Loss::BPR : Loss::Hinge;
        optimizer_ = uniform(0.0, 1.0) < 0.5 ? Optimizer::Adam : Optimizer::Adagrad;
    }

    std::size_t num_items_;
    std::size_t max_sequence_length_;
    std::size_t item_embedding_dim_ = 16;
    float learning_rate_ = 0.01f;
    float l2_penalty_ = 0.0f;
    Loss loss_ = Loss::BPR;
    Optimizer optimizer_ = Optimizer::Adam;
    Parallelism parallelism_ = Parallelism::Synchronous;
    XorShiftRng rng_;
    std::size_t num_threads_ = 1;
    std::size_t num_epochs_ = 10;
    std::size_t batch_sequences_ = 32;
    bool partition_item_table_ = false;
};

} // namespace detail

// ---- sbr::models::lstm (lstm.rs) ---------------------------------------------------------------
namespace lstm {

/// Type of LSTM layer (lstm.rs:28-35).
enum class LSTMVariant { Normal, Coupled };

/// An LSTM-based sequence model for implicit feedback (lstm.rs:386-416).
class ImplicitLSTMModel : public detail::ImplicitSequenceModel {
    using detail::ImplicitSequenceModel::ImplicitSequenceModel;

  public:
    /// A model written by `save` (≙ deserialising the serde-derived ImplicitLSTMModel, lstm.rs:386).
    static ImplicitLSTMModel load(const std::string& path) { return ImplicitLSTMModel(path, 0); }
};

/// Hyperparameters for the ImplicitLSTMModel (lstm.rs:39-202).
class Hyperparameters : public detail::HyperparametersBase<Hyperparameters> {
  public:
    /// Build new hyperparameters.
    static Hyperparameters new_(std::size_t num_items, std::size_t max_sequence_length) {
        return Hyperparameters(num_items, max_sequence_length);
    }
    Hyperparameters(std::size_t num_items, std::size_t max_sequence_length)
        : detail::HyperparametersBase<Hyperparameters>(num_items, max_sequence_length) {}
    /// Set the LSTM variant.
    Hyperparameters& lstm_variant(LSTMVariant v) { lstm_type_ = v; return *this; }
    /// Generate a random point of the search space (lstm.rs:141-172).
    static Hyperparameters random(std::size_t num_items, XorShiftRng& rng) {
        Hyperparameters h(num_items, 16);
        h.randomize_common(rng);
        auto uniform = [&]() { return rng.unit(); };
        h.lstm_type_ = uniform() < 0.5 ? LSTMVariant::Normal : LSTMVariant::Coupled;
        h.parallelism_ = uniform() < 0.5 ? Parallelism::Asynchronous : Parallelism::Synchronous;
        h.num_epochs_ = (std::size_t)1 << (3 + rng.below(4));
        return h;
    }
    /// Build a model out of the chosen hyperparameters: parameters are initialised on the device
    /// from the builder's RNG (lstm.rs:174-201).
    ImplicitLSTMModel build() const {
        return ImplicitLSTMModel(hparams(lstm_type_ == LSTMVariant::Normal ? SBR_MODEL_LSTM_NORMAL : SBR_MODEL_LSTM_COUPLED),
                                 partition_item_table_);
    }

  private:
    LSTMVariant lstm_type_ = LSTMVariant::Coupled;
};

} // namespace lstm

// ---- sbr::models::ewma (ewma.rs) ---------------------------------------------------------------
namespace ewma {

/// Implicit EWMA model (ewma.rs:401-429).  State recurrence as coded at ewma.rs:302-313.
class ImplicitEWMAModel : public detail::ImplicitSequenceModel {
    using detail::ImplicitSequenceModel::ImplicitSequenceModel;

  public:
    /// A model written by `save` (≙ deserialising the serde-derived ImplicitEWMAModel, ewma.rs:401).
    static ImplicitEWMAModel load(const std::string& path) { return ImplicitEWMAModel(path, 1); }
};

/// Hyperparameters describing the EWMA model (ewma.rs:45-206).
class Hyperparameters : public detail::HyperparametersBase<Hyperparameters> {
  public:
    static Hyperparameters new_(std::size_t num_items, std::size_t max_sequence_length) {
        return Hyperparameters(num_items, max_sequence_length);
    }
    Hyperparameters(std::size_t num_items, std::size_t max_sequence_length)
        : detail::HyperparametersBase<Hyperparameters>(num_items, max_sequence_length) {}
    static Hyperparameters random(std::size_t num_items, XorShiftRng& rng) {
        Hyperparameters h(num_items, 16);
        h.randomize_common(rng);
        h.parallelism_ = rng.unit() < 0.5 ? Parallelism::Asynchronous : Parallelism::Synchronous;
        h.num_epochs_ = (std::size_t)1 << (3 + rng.below(4));
        return h;
    }
    /// Build the implicit EWMA model (ewma.rs:201-205).
    ImplicitEWMAModel build() const { return ImplicitEWMAModel(hparams(SBR_MODEL_EWMA), partition_item_table_); }
};

} // namespace ewma
} // namespace models

// =============================================================================================
// sbr::evaluation  (src/evaluation.rs)
// =============================================================================================
namespace evaluation {

/// MRR of the last item of every test sequence, the items before it being the inputs
/// (evaluation.rs:12-48), through the model's OnlineRankingModel interface — the reference's loop:
/// one user_representation + one full-catalogue predict per user, history masked to f32::MIN,
/// rank = #{score >= score[test item]}.
template <class Model>
Result<float, PredictionError> mrr_score_generic(const Model& model, const data::CompressedInteractions& test) {
    std::vector<ItemId> item_ids(test.num_items());
    std::iota(item_ids.begin(), item_ids.end(), (ItemId)0);
    float sum = 0.0f;
    std::size_t count = 0;
    for (const auto& user : test.iter_users()) {
        if (user.len() < 2) continue;
        const std::vector<ItemId> train_items(user.item_ids, user.item_ids + user.len() - 1);
        const ItemId test_item = user.item_ids[user.len() - 1];
        auto representation = model.user_representation(train_items);
        if (representation.is_err()) return Result<float, PredictionError>::Err(representation.unwrap_err());
        auto scored = model.predict(representation.unwrap(), item_ids);
        if (scored.is_err()) return Result<float, PredictionError>::Err(scored.unwrap_err());
        std::vector<float>& predictions = scored.unwrap();
        for (ItemId seen : train_items) predictions[seen] = std::numeric_limits<float>::lowest();
        const float test_score = predictions[test_item];
        std::size_t rank = 0;
        for (float p : predictions) rank += p >= test_score;
        sum += 1.0f / (float)rank;
        count += 1;
    }
    return Result<float, PredictionError>::Ok(sum / (float)count);
}

/// mrr_score for the engine's models: the whole evaluation in one device call (sbr_mrr_score:
/// batched user representations, MFMA score GEMM with the rank count in its epilogue).  Integer
/// ranks are identical to mrr_score_generic's.
inline Result<float, PredictionError> mrr_score(const models::detail::ImplicitSequenceModel& model,
                                                const data::CompressedInteractions& test,
                                                std::vector<std::uint32_t>* out_ranks = nullptr) {
    float mrr = 0.0f;
    std::uint64_t ranked = 0;
    std::vector<std::uint32_t> ranks(test.num_users());
    const sbr_status st = sbr_mrr_score(model.handle(), test.user_pointers().data(), test.item_ids().data(),
                                        (std::uint64_t)test.num_users(), &mrr, ranks.data(), &ranked);
    if (st == SBR_ERR_INVALID_PREDICTION) return Result<float, PredictionError>::Err(PredictionError::InvalidPredictionValue);
    models::detail::check(st, "sbr_mrr_score");
    if (out_ranks) {
        ranks.resize(ranked);
        *out_ranks = std::move(ranks);
    }
    return Result<float, PredictionError>::Ok(mrr);
}

} // namespace evaluation

// =============================================================================================
// sbr::datasets  (src/datasets.rs) — the part that needs no network
// =============================================================================================
namespace datasets {

/// Reads the MovieLens CSV the reference downloads (datasets.rs:57-60: header
/// `user_id,item_id,rating,timestamp`, deserialised into Interaction{user_id,item_id,timestamp}).
inline data::Interactions read_csv(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error(path + ": cannot open");
    std::string line;
    if (!std::getline(f, line)) throw std::runtime_error(path + ": empty file");
    std::vector<std::string> header;
    {
        std::stringstream ss(line);
        std::string cell;
        while (std::getline(ss, cell, ',')) {
            while (!cell.empty() && (cell.back() == '\r' || cell.back() == ' ')) cell.pop_back();
            header.push_back(cell);
        }
    }
    auto column = [&](const char* name) {
        const auto it = std::find(header.begin(), header.end(), name);
        if (it == header.end()) throw std::runtime_error(path + ": no column " + name);
        return (std::size_t)(it - header.begin());
    };
    const std::size_t cu = column("user_id"), ci = column("item_id"), ct = column("timestamp");
    std::vector<data::Interaction> rows;
    while (std::getline(f, line)) {
        if (line.empty()) continue;
        std::vector<std::string> cells;
        std::stringstream ss(line);
        std::string cell;
        while (std::getline(ss, cell, ',')) cells.push_back(cell);
        if (cells.size() <= std::max(cu, std::max(ci, ct))) throw std::runtime_error(path + ": short row");
        rows.emplace_back((UserId)std::stoull(cells[cu]), (ItemId)std::stoull(cells[ci]), (Timestamp)std::stoull(cells[ct]));
    }
    return data::Interactions::from(std::move(rows));
}

/// Same name as the reference entry point (datasets.rs:66-71); reads a local copy (argument, or
/// $SBR_MOVIELENS_PATH) instead of downloading — there is no egress on the GPU hosts.
inline data::Interactions download_movielens_100k(const std::string& path = "") {
    std::string p = path;
    if (p.empty())
        if (const char* env = std::getenv("SBR_MOVIELENS_PATH")) p = env;
    if (p.empty()) throw std::runtime_error("download_movielens_100k: no network here; set SBR_MOVIELENS_PATH to data.csv");
    return read_csv(p);
}

} // namespace datasets
} // namespace sbr

#endif // SBR_HPP
