// libzeth/gpu: the C++ host side above the C ABI of zeth_gpu.h -- Zeth's proving entry points with the reference's
// names, argument meaning and error behaviour, computing on the MI355X through libzeth_gpu.so.
//
//   groth16_snark<ppT>::generate_proof(pk, primary, auxiliary)      libzeth/snarks/groth16/groth16_snark.hpp:37-40, .tcc:36-46
//   r1cs_gg_ppzksnark_prover(pk, primary, auxiliary, force_pow_2)   called at groth16_snark.tcc:44-45
//   groth16_snark<ppT>::proving_key_{read,write}_bytes, proof_*     groth16_snark.tcc:93-175
//   is_well_formed<ppT>(pk)                                         groth16_snark.tcc:193-213
//   multi_exp<ppT, GroupT>(gs, fs) and the iterator overload        libzeth/core/multi_exp.hpp:13-22, .tcc:13-35
//   basic_radix2_domain<ppT>::{FFT, iFFT, cosetFFT, icosetFFT}      libfqfft, called at libzeth/core/evaluator_from_lagrange.tcc:43
//   evaluator_from_lagrange<ppT, GroupT>                            libzeth/core/evaluator_from_lagrange.tcc:14-45
//
// There is no CPU path behind any of these: without a usable HIP device every call throws std::runtime_error.
#ifndef LIBZETH_GPU_GROTH16_SNARK_HPP
#define LIBZETH_GPU_GROTH16_SNARK_HPP

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <random>

#include "serialization.hpp"
#include "types.hpp"

namespace libzeth_gpu
{

/// One zg_ctx per HIP device, created on first use and shared by the stateless entry points below.  Calls on one
/// context must not overlap (zeth_gpu.h), so every entry point holds the context's mutex for its duration -- the
/// one-at-a-time discipline prover_server's shared circuit_wrapper already needs.
class device_context
{
public:
    /// shared ownership: every object that holds device state (keys, bases) keeps its context alive, so the order in
    /// which statics are destroyed at process exit cannot free a context under a key that still needs it
    static std::shared_ptr<device_context> get_shared(int device = 0)
    {
        static std::mutex registry_mutex;
        static std::map<int, std::shared_ptr<device_context>> registry;
        std::lock_guard<std::mutex> lock(registry_mutex);
        std::shared_ptr<device_context> &slot = registry[device];
        if (!slot) {
            slot.reset(new device_context(device));
        }
        return slot;
    }
    static device_context &get(int device = 0) { return *get_shared(device); }
    ~device_context() { zg_ctx_destroy(ctx_); }
    zg_ctx *ctx() const { return ctx_; }
    std::mutex &mutex() { return mutex_; }

private:
    explicit device_context(int device) : ctx_(nullptr) { check(zg_ctx_create(device, &ctx_)); }
    device_context(const device_context &) = delete;
    zg_ctx *ctx_;
    std::mutex mutex_;
};

/// The proving key resident on the device: bases in HBM, constraint matrices as CSR, per-proof scratch.  The
/// analogue of loading the keypair once before serving proofs (prover_server/prover_server.cpp:449-493,
/// zeth_tool/prove_cmd.cpp:39-43).
///
/// One object = one prover: calls on it are serialised by its own mutex.  clone() gives ANOTHER prover over the same
/// resident key (zg_pk_clone: shared bases and matrices, own scratch, streams and context), so that prove() on the
/// clone overlaps prove() on the original -- how several proofs are kept in flight on one device.  Clones keep the
/// uploaded key alive: the original may be destroyed first.
template<typename ppT> class device_proving_key
{
public:
    device_proving_key(const r1cs_gg_ppzksnark_proving_key<ppT> &pk, int device = 0) : root_(new root(device)), handle_(nullptr), call_mutex_(&root_->dc.mutex())
    {
        using FieldT = Fr<ppT>;
        const r1cs_constraint_system<FieldT> &cs = pk.constraint_system;
        const size_t nc = cs.num_constraints();
        // three CSR matrices over (constraint, variable index)
        std::vector<uint64_t> row_ptr[3];
        std::vector<uint32_t> col[3];
        std::vector<FieldT> val[3];
        for (int k = 0; k < 3; k++) {
            row_ptr[k].reserve(nc + 1);
            row_ptr[k].push_back(0);
        }
        for (const r1cs_constraint<FieldT> &c : cs.constraints) {
            const linear_combination<FieldT> *lcs[3] = {&c.a, &c.b, &c.c};
            for (int k = 0; k < 3; k++) {
                for (const linear_term<FieldT> &t : lcs[k]->terms) {
                    if (t.index > cs.num_variables()) {
                        throw std::invalid_argument("constraint system: variable index out of range");
                    }
                    col[k].push_back((uint32_t)t.index);
                    val[k].push_back(t.coeff);
                }
                row_ptr[k].push_back(col[k].size());
            }
        }
        if (pk.B_query.indices.size() != pk.B_query.values.size()) {
            throw std::invalid_argument("proving key: B_query indices / values size mismatch");
        }
        const size_t nb = pk.B_query.indices.size();
        std::vector<uint32_t> b_idx(nb);
        G2_vector<ppT> b_g2(nb);
        G1_vector<ppT> b_g1(nb);
        for (size_t i = 0; i < nb; i++) {
            if (pk.B_query.indices[i] > cs.num_variables()) {
                throw std::invalid_argument("proving key: B_query index out of range");
            }
            b_idx[i] = (uint32_t)pk.B_query.indices[i];
            b_g2[i] = pk.B_query.values[i].g;
            b_g1[i] = pk.B_query.values[i].h;
        }
        zg_pk_view v;
        memset(&v, 0, sizeof(v));
        v.num_inputs = cs.num_inputs();
        v.num_variables = cs.num_variables();
        v.num_constraints = nc;
        zg_r1cs_matrix *mats[3] = {&v.a, &v.b, &v.c};
        for (int k = 0; k < 3; k++) {
            mats[k]->row_ptr = row_ptr[k].data();
            mats[k]->col = col[k].data();
            mats[k]->val = val[k].data();
        }
        v.alpha_g1 = &pk.alpha_g1;
        v.beta_g1 = &pk.beta_g1;
        v.beta_g2 = &pk.beta_g2;
        v.delta_g1 = &pk.delta_g1;
        v.delta_g2 = &pk.delta_g2;
        v.A_query = pk.A_query.data();
        v.A_len = pk.A_query.size();
        v.B_indices = b_idx.data();
        v.B_g2 = b_g2.data();
        v.B_g1 = b_g1.data();
        v.B_len = nb;
        v.H_query = pk.H_query.data();
        v.H_len = pk.H_query.size();
        v.L_query = pk.L_query.data();
        v.L_len = pk.L_query.size();
        v.form = ZG_FORM_PLAIN;
        std::lock_guard<std::mutex> lock(root_->dc.mutex());
        check(zg_pk_create(root_->dc.ctx(), ppT::curve_id, &v, &root_->handle));
        handle_ = root_->handle;
    }
    ~device_proving_key()
    {
        if (handle_ && handle_ != root_->handle) {
            zg_pk_free(handle_); // a clone: its own context, nothing shared with other callers
        }
        // the uploaded key itself goes with the last object that refers to it (root's destructor)
    }
    device_proving_key(const device_proving_key &) = delete;
    device_proving_key &operator=(const device_proving_key &) = delete;

    /// A second prover over the same resident key, for another calling thread (zg_pk_clone)
    std::unique_ptr<device_proving_key> clone() const
    {
        std::unique_ptr<device_proving_key> c(new device_proving_key(root_));
        std::lock_guard<std::mutex> lock(root_->dc.mutex()); // the original must not be proving while it is cloned
        check(zg_pk_clone(root_->handle, &c->handle_));
        return c;
    }
    bool is_clone() const { return handle_ != root_->handle; }

    size_t domain_size() const
    {
        size_t m = 0;
        check(zg_pk_domain_size(handle_, &m));
        return m;
    }

    /// The prover with its randomness passed in (r, s plain): reproducible proofs for tests.
    r1cs_gg_ppzksnark_proof<ppT> prove(
        const r1cs_primary_input<Fr<ppT>> &primary, const r1cs_auxiliary_input<Fr<ppT>> &auxiliary, const Fr<ppT> &r, const Fr<ppT> &s) const
    {
        r1cs_gg_ppzksnark_proof<ppT> proof;
        static_assert(sizeof(proof) == 2 * sizeof(G1<ppT>) + sizeof(G2<ppT>), "proof must be g_A || g_B || g_C without padding");
        std::lock_guard<std::mutex> lock(*call_mutex_);
        check(zg_prove(
            handle_, primary.data(), primary.size(), auxiliary.data(), auxiliary.size(), ZG_FORM_PLAIN, r.data, s.data, ZG_FORM_PLAIN, &proof));
        return proof;
    }

    /// libsnark::r1cs_to_qap_witness_map(...).coefficients_for_H (domain size + 1 values)
    std::vector<Fr<ppT>> witness_map(const r1cs_primary_input<Fr<ppT>> &primary, const r1cs_auxiliary_input<Fr<ppT>> &auxiliary) const
    {
        std::vector<Fr<ppT>> h(domain_size() + 1);
        std::lock_guard<std::mutex> lock(*call_mutex_);
        check(zg_witness_map(
            handle_, primary.data(), primary.size(), auxiliary.data(), auxiliary.size(), ZG_FORM_PLAIN, ZG_FORM_PLAIN, h.data()));
        return h;
    }

private:
    // the uploaded key: lives on the device's shared context (whose mutex serialises every call on that context) and
    // is freed with the last prover that refers to it
    struct root {
        std::shared_ptr<device_context> dcp; // keeps the context alive for as long as the key
        device_context &dc;
        zg_pk *handle;
        explicit root(int device) : dcp(device_context::get_shared(device)), dc(*dcp), handle(nullptr) {}
        ~root()
        {
            if (handle) {
                std::lock_guard<std::mutex> lock(dc.mutex());
                zg_pk_free(handle);
            }
        }
    };
    explicit device_proving_key(const std::shared_ptr<root> &r) : root_(r), handle_(nullptr), call_mutex_(&own_mutex_) {}
    // declared first, destroyed last: a clone's handle is freed (destructor body) before the key it was cloned from
    std::shared_ptr<root> root_;
    zg_pk *handle_;
    mutable std::mutex own_mutex_; // a clone has a context of its own: only calls on THIS object exclude each other
    std::mutex *call_mutex_;       // the original computes on the device's shared context: that context's mutex
};

/// Fr::random_element(): uniform in [0, r) from the OS entropy source (libsnark draws the prover's r and s this
/// way, inside the prover -- SURVEY 8b).
template<typename ppT> Fr<ppT> random_field_element()
{
    std::random_device rd;
    const uint64_t top_mask = (ppT::r_bits % 64) ? ((uint64_t(1) << (ppT::r_bits % 64)) - 1) : ~uint64_t(0);
    for (;;) {
        Fr<ppT> x;
        for (size_t i = 0; i < 4; i++) {
            x.data[i] = ((uint64_t)rd() << 32) | (uint64_t)rd();
        }
        x.data[3] &= top_mask;
        if (x.less_than(ppT::r)) {
            return x;
        }
    }
}

/// Proofs in flight across devices and within each device: one resident copy of the proving key per device and up to
/// `in_flight_per_device` provers over it (the original + zg_pk_clone clones, created when first needed).  prove() takes
/// a free prover -- on the device with the fewest proofs in flight -- and blocks only when every prover of every device
/// is busy.  This is what prover_server's concurrent handlers (prover_server/prover_server.cpp:118-201; one shared
/// keypair, :51,451) call into: K handler threads keep K proofs in flight (BASELINE configs[4]).  Independent proofs are
/// replicas -- no exchange between devices.  One process, N devices.
template<typename ppT> class device_proving_key_pool
{
public:
    // measured (JoinSplit-shaped proof, one MI355X): 2 / 3 / 4 / 5 / 6 / 8 in flight = 390 / 443 / 439 / 363 / 335 / 341 proofs/s
    // (from five on the callers' and lanes' host threads outnumber a GPU's share of cores)
    static constexpr size_t default_in_flight_per_device = 4;

    device_proving_key_pool(
        const r1cs_gg_ppzksnark_proving_key<ppT> &pk, const std::vector<int> &devices, size_t in_flight_per_device = default_in_flight_per_device)
        : max_per_device_(in_flight_per_device ? in_flight_per_device : 1), in_flight_(0), peak_in_flight_(0), proofs_(0)
    {
        if (devices.empty()) {
            throw std::invalid_argument("device_proving_key_pool: no devices");
        }
        for (int d : devices) {
            std::unique_ptr<device_state> st(new device_state());
            st->slots.emplace_back(new slot{std::unique_ptr<device_proving_key<ppT>>(new device_proving_key<ppT>(pk, d)), false});
            devices_.push_back(std::move(st));
        }
    }
    size_t size() const { return devices_.size(); }
    size_t in_flight_per_device() const { return max_per_device_; }
    /// the largest number of proofs that were in flight at once, and the proofs completed, since construction
    size_t peak_in_flight() const
    {
        std::lock_guard<std::mutex> lock(m_);
        return peak_in_flight_;
    }
    size_t proofs_completed() const
    {
        std::lock_guard<std::mutex> lock(m_);
        return proofs_;
    }

    r1cs_gg_ppzksnark_proof<ppT> prove(
        const r1cs_primary_input<Fr<ppT>> &primary, const r1cs_auxiliary_input<Fr<ppT>> &auxiliary, const Fr<ppT> &r, const Fr<ppT> &s)
    {
        lease l(*this);
        return l.prover().prove(primary, auxiliary, r, s);
    }
    r1cs_gg_ppzksnark_proof<ppT> prove(const r1cs_primary_input<Fr<ppT>> &primary, const r1cs_auxiliary_input<Fr<ppT>> &auxiliary)
    {
        return prove(primary, auxiliary, random_field_element<ppT>(), random_field_element<ppT>());
    }

    /// Creates every prover of every device now and runs one proof of the given assignment on each, so that no later
    /// caller pays for a clone (its streams, scratch and page-locked staging: tens of milliseconds) or for a prover's
    /// first proof.  Optional -- provers are otherwise created when first needed; a server calls it once before it
    /// starts to serve.  Waits for proofs in flight to finish.
    void warm_up(const r1cs_primary_input<Fr<ppT>> &primary, const r1cs_auxiliary_input<Fr<ppT>> &auxiliary)
    {
        std::lock_guard<std::mutex> one_at_a_time(warm_up_m_); // two of these at once would each hold half the provers
        std::vector<std::unique_ptr<lease>> held;
        for (size_t i = 0; i < devices_.size() * max_per_device_; i++) {
            held.emplace_back(new lease(*this)); // this thread holds the others: every lease is a different prover
        }
        const Fr<ppT> r = random_field_element<ppT>(), s = random_field_element<ppT>();
        for (auto &l : held) {
            (void)l->prover().prove(primary, auxiliary, r, s);
        }
    }

private:
    struct slot {
        std::unique_ptr<device_proving_key<ppT>> prover;
        bool busy;
    };
    struct device_state {
        std::vector<std::unique_ptr<slot>> slots; // slots[0] holds the uploaded key, the others are its clones
        size_t busy = 0;
        size_t growing = 0; // clones being created (outside the pool's lock)
    };
    // one proof's hold on a prover; released on every exit path
    class lease
    {
    public:
        explicit lease(device_proving_key_pool &pool) : pool_(pool), dev_(nullptr), slot_(nullptr)
        {
            std::unique_lock<std::mutex> lock(pool_.m_);
            for (;;) {
                // the device with the fewest proofs in flight that still has, or may still create, a free prover
                device_state *best = nullptr;
                for (auto &d : pool_.devices_) {
                    const bool has_free = d->busy < d->slots.size();
                    const bool can_grow = d->slots.size() + d->growing < pool_.max_per_device_;
                    if ((has_free || can_grow) && (!best || d->busy + d->growing < best->busy + best->growing)) {
                        best = d.get();
                    }
                }
                if (!best) {
                    pool_.cv_.wait(lock);
                    continue;
                }
                dev_ = best;
                for (auto &sl : best->slots) {
                    if (!sl->busy) {
                        slot_ = sl.get();
                        break;
                    }
                }
                if (!slot_) {
                    // every prover of this device is busy and it may have more: clone the key's prover without holding
                    // the pool (a clone allocates its scratch and streams: milliseconds)
                    best->growing++;
                    lock.unlock();
                    std::unique_ptr<slot> fresh;
                    try {
                        fresh.reset(new slot{best->slots[0]->prover->clone(), true});
                    } catch (...) {
                        lock.lock();
                        best->growing--;
                        pool_.cv_.notify_all();
                        throw;
                    }
                    lock.lock();
                    best->growing--;
                    slot_ = fresh.get();
                    best->slots.push_back(std::move(fresh));
                }
                slot_->busy = true;
                best->busy++;
                pool_.in_flight_++;
                if (pool_.in_flight_ > pool_.peak_in_flight_) {
                    pool_.peak_in_flight_ = pool_.in_flight_;
                }
                return;
            }
        }
        ~lease()
        {
            std::lock_guard<std::mutex> lock(pool_.m_);
            slot_->busy = false;
            dev_->busy--;
            pool_.in_flight_--;
            pool_.proofs_++;
            pool_.cv_.notify_one();
        }
        lease(const lease &) = delete;
        const device_proving_key<ppT> &prover() const { return *slot_->prover; }

    private:
        device_proving_key_pool &pool_;
        device_state *dev_;
        slot *slot_;
    };

    std::vector<std::unique_ptr<device_state>> devices_;
    const size_t max_per_device_;
    mutable std::mutex m_;
    std::mutex warm_up_m_;
    std::condition_variable cv_;
    size_t in_flight_, peak_in_flight_, proofs_;
};

namespace internal
{
// Device keys built by the stateless entry points, by host key address: each a pool of provers over one uploaded key
// (device 0; up to stateless_in_flight() proofs of concurrent callers overlap, as prover_server's handlers need).  A fingerprint over the key's sizes, its
// single elements, samples of all four queries (B included) and of the constraint system guards against a different
// or modified key at the same address.  At most MAX_KEYS device keys are kept (least recently used goes first: a
// device key can hold 12-13 times its H query); release_device_keys() drops them all, and runs by itself at exit while
// the HIP runtime is still up.
template<typename ppT> struct device_key_cache {
    static constexpr size_t MAX_KEYS = 4;
    struct entry {
        uint64_t fingerprint;
        uint64_t last_use;
        std::shared_ptr<device_proving_key_pool<ppT>> key;
    };
    struct state {
        std::mutex m;
        std::map<const void *, entry> cache;
        uint64_t clock = 0;
        bool exit_hook = false;
    };
    static state &st()
    {
        static state *s = new state(); // never destroyed: nothing of it runs inside static destructors
        return *s;
    }
    template<typename T> static void mix_samples(uint64_t &h, const std::vector<T> &v)
    {
        auto mix = [&h](const void *p, size_t n) {
            const uint8_t *b = (const uint8_t *)p;
            for (size_t i = 0; i < n; i++) {
                h = (h ^ b[i]) * 0x100000001b3ull;
            }
        };
        const size_t n = v.size();
        mix(&n, sizeof(n));
        const size_t step = n / 61 + 1; // up to ~61 evenly spaced elements, always the first and the last
        for (size_t i = 0; i < n; i += step) {
            mix(&v[i], sizeof(T));
        }
        if (n) {
            mix(&v[n - 1], sizeof(T));
        }
    }
    static uint64_t fingerprint(const r1cs_gg_ppzksnark_proving_key<ppT> &pk)
    {
        uint64_t h = 0xcbf29ce484222325ull;
        auto mix = [&h](const void *p, size_t n) {
            const uint8_t *b = (const uint8_t *)p;
            for (size_t i = 0; i < n; i++) {
                h = (h ^ b[i]) * 0x100000001b3ull;
            }
        };
        mix(&pk.alpha_g1, sizeof(pk.alpha_g1));
        mix(&pk.beta_g1, sizeof(pk.beta_g1));
        mix(&pk.beta_g2, sizeof(pk.beta_g2));
        mix(&pk.delta_g1, sizeof(pk.delta_g1));
        mix(&pk.delta_g2, sizeof(pk.delta_g2));
        mix_samples(h, pk.A_query);
        mix_samples(h, pk.B_query.indices);
        mix_samples(h, pk.B_query.values);
        mix_samples(h, pk.H_query);
        mix_samples(h, pk.L_query);
        const auto &cs = pk.constraint_system;
        const size_t sizes[3] = {cs.num_constraints(), cs.num_variables(), cs.num_inputs()};
        mix(sizes, sizeof(sizes));
        const size_t nc = cs.constraints.size(), step = nc / 61 + 1;
        for (size_t i = 0; i < nc; i += step) {
            const linear_combination<Fr<ppT>> *lcs[3] = {&cs.constraints[i].a, &cs.constraints[i].b, &cs.constraints[i].c};
            for (const auto *lc : lcs) {
                for (const auto &t : lc->terms) {
                    mix(&t.index, sizeof(t.index));
                    mix(&t.coeff, sizeof(t.coeff));
                }
            }
        }
        return h;
    }
    static void release_all()
    {
        state &s = st();
        std::lock_guard<std::mutex> lock(s.m);
        s.cache.clear();
    }
    static std::shared_ptr<device_proving_key_pool<ppT>> lookup(const r1cs_gg_ppzksnark_proving_key<ppT> &pk)
    {
        state &s = st();
        std::lock_guard<std::mutex> lock(s.m);
        const uint64_t fp = fingerprint(pk);
        auto it = s.cache.find(&pk);
        if (it != s.cache.end() && it->second.fingerprint == fp) {
            it->second.last_use = ++s.clock;
            return it->second.key;
        }
        if (it != s.cache.end()) {
            s.cache.erase(it); // the key at this address changed: its device copy is stale
        }
        while (s.cache.size() >= MAX_KEYS) {
            auto oldest = s.cache.begin();
            for (auto jt = s.cache.begin(); jt != s.cache.end(); ++jt) {
                if (jt->second.last_use < oldest->second.last_use) {
                    oldest = jt;
                }
            }
            s.cache.erase(oldest);
        }
        entry e{fp, ++s.clock, std::make_shared<device_proving_key_pool<ppT>>(pk, std::vector<int>{0})};
        s.cache[&pk] = e;
        if (!s.exit_hook) {
            // registered after the first key exists, hence after the HIP runtime came up: at exit this runs before the
            // runtime's own teardown and frees the device keys while their contexts still work
            s.exit_hook = true;
            std::atexit(&device_key_cache<ppT>::release_all);
        }
        return e.key;
    }
};
} // namespace internal

/// Drops every device key the stateless generate_proof / r1cs_gg_ppzksnark_prover keep (they are re-uploaded on the
/// next use).  For hosts that want their HBM back, or that unload the library before exit.
template<typename ppT> void release_device_keys() { internal::device_key_cache<ppT>::release_all(); }

/// libsnark::r1cs_gg_ppzksnark_prover as Zeth calls it.  The key is uploaded on first use and stays resident for the
/// process lifetime.  Concurrent calls with one key overlap on the device (up to
/// device_proving_key_pool::default_in_flight_per_device proofs in flight; further callers wait for a free prover).  Zeth always passes force_pow_2_domain = true (groth16_snark.tcc:43-45); the device transforms
/// are radix-2 only, so false is rejected.
template<typename ppT>
r1cs_gg_ppzksnark_proof<ppT> r1cs_gg_ppzksnark_prover(
    const r1cs_gg_ppzksnark_proving_key<ppT> &pk,
    const r1cs_primary_input<Fr<ppT>> &primary_input,
    const r1cs_auxiliary_input<Fr<ppT>> &auxiliary_input,
    bool force_pow_2_domain)
{
    if (!force_pow_2_domain) {
        throw std::invalid_argument("r1cs_gg_ppzksnark_prover: only power-of-two domains are supported");
    }
    const std::shared_ptr<device_proving_key_pool<ppT>> provers = internal::device_key_cache<ppT>::lookup(pk);
    return provers->prove(primary_input, auxiliary_input, random_field_element<ppT>(), random_field_element<ppT>());
}

template<typename ppT> bool container_is_well_formed(const G1_vector<ppT> &points)
{
    size_t bad = 0;
    check(zg_host_points_check(ppT::curve_id, ZG_GROUP_G1, points.data(), points.size(), ZG_FORM_PLAIN, &bad));
    return bad == points.size();
}

template<typename ppT> bool group_element_is_well_formed(const G1<ppT> &p)
{
    size_t bad = 0;
    check(zg_host_points_check(ppT::curve_id, ZG_GROUP_G1, &p, 1, ZG_FORM_PLAIN, &bad));
    return bad == 1;
}

template<typename ppT> bool group_element_is_well_formed(const G2<ppT> &p)
{
    size_t bad = 0;
    check(zg_host_points_check(ppT::curve_id, ZG_GROUP_G2, &p, 1, ZG_FORM_PLAIN, &bad));
    return bad == 1;
}

/// groth16_snark.tcc:193-213: alpha, beta, delta, A_query, L_query and every B_query entry are on their curves
/// (like the reference, H_query is not checked)
template<typename ppT> bool is_well_formed(const r1cs_gg_ppzksnark_proving_key<ppT> &pk)
{
    if (!group_element_is_well_formed<ppT>(pk.alpha_g1) || !group_element_is_well_formed<ppT>(pk.beta_g1) ||
        !group_element_is_well_formed<ppT>(pk.beta_g2) || !group_element_is_well_formed<ppT>(pk.delta_g1) ||
        !group_element_is_well_formed<ppT>(pk.delta_g2) || !container_is_well_formed<ppT>(pk.A_query) ||
        !container_is_well_formed<ppT>(pk.L_query)) {
        return false;
    }
    for (const auto &b : pk.B_query.values) {
        if (!group_element_is_well_formed<ppT>(b.g) || !group_element_is_well_formed<ppT>(b.h)) {
            return false;
        }
    }
    return true;
}

/// groth16_snark.tcc:215-224
template<typename ppT> bool is_well_formed(const r1cs_gg_ppzksnark_verification_key<ppT> &vk)
{
    if (!group_element_is_well_formed<ppT>(vk.alpha_g1) || !group_element_is_well_formed<ppT>(vk.beta_g2) ||
        !group_element_is_well_formed<ppT>(vk.delta_g2) || !group_element_is_well_formed<ppT>(vk.ABC_g1.first)) {
        return false;
    }
    return container_is_well_formed<ppT>(vk.ABC_g1.rest.values);
}

/// The proving half of libzeth::groth16_snark<ppT> (groth16_snark.hpp:15-79) and the key / proof codecs its callers use
/// (prover_server.cpp:462,475,482,490; prove_cmd.cpp:39-62).  generate_setup is in gpu_groth16_generator.hpp;
/// verification stays with libsnark on the CPU: it is outside the accelerated path.
template<typename ppT> class groth16_snark
{
public:
    using proving_key = r1cs_gg_ppzksnark_proving_key<ppT>;
    using verification_key = r1cs_gg_ppzksnark_verification_key<ppT>;
    using keypair = r1cs_gg_ppzksnark_keypair<ppT>;
    using proof = r1cs_gg_ppzksnark_proof<ppT>;

    static constexpr const char *name = "GROTH16";

    /// generate_proof(pk, pb) (groth16_snark.hpp:32-35, .tcc:27-34): the protoboard's own split of its assignment
    static proof generate_proof(const proving_key &proving_key, const protoboard<Fr<ppT>> &pb)
    {
        return generate_proof(proving_key, pb.primary_input(), pb.auxiliary_input());
    }

    static proof generate_proof(
        const proving_key &proving_key, const r1cs_primary_input<Fr<ppT>> &primary_input, const r1cs_auxiliary_input<Fr<ppT>> auxiliary_input)
    {
        // "For now, force a pow2 domain, in case the key came from the MPC" (groth16_snark.tcc:42-45)
        return r1cs_gg_ppzksnark_prover<ppT>(proving_key, primary_input, auxiliary_input, true);
    }

    static void proving_key_write_bytes(const proving_key &pk, std::ostream &out_s)
    {
        if (!is_well_formed<ppT>(pk)) {
            throw std::invalid_argument("proving key (write) not well-formed");
        }
        group_element_write_bytes(pk.alpha_g1, out_s);
        group_element_write_bytes(pk.beta_g1, out_s);
        group_element_write_bytes(pk.beta_g2, out_s);
        group_element_write_bytes(pk.delta_g1, out_s);
        group_element_write_bytes(pk.delta_g2, out_s);
        group_elements_write_bytes(pk.A_query, out_s);
        knowledge_commitment_vector_write_bytes(pk.B_query, out_s);
        group_elements_write_bytes(pk.H_query, out_s);
        group_elements_write_bytes(pk.L_query, out_s);
        r1cs_write_bytes(pk.constraint_system, out_s);
    }

    static void proving_key_read_bytes(proving_key &pk, std::istream &in_s)
    {
        group_element_read_bytes(pk.alpha_g1, in_s);
        group_element_read_bytes(pk.beta_g1, in_s);
        group_element_read_bytes(pk.beta_g2, in_s);
        group_element_read_bytes(pk.delta_g1, in_s);
        group_element_read_bytes(pk.delta_g2, in_s);
        group_elements_read_bytes(pk.A_query, in_s);
        knowledge_commitment_vector_read_bytes(pk.B_query, in_s);
        group_elements_read_bytes(pk.H_query, in_s);
        group_elements_read_bytes(pk.L_query, in_s);
        r1cs_read_bytes(pk.constraint_system, in_s);
        if (!is_well_formed<ppT>(pk)) {
            throw std::invalid_argument("proving key (read) not well-formed");
        }
    }

    /// groth16_snark.tcc:76-91
    static void verification_key_write_bytes(const verification_key &vk, std::ostream &out_s)
    {
        if (!is_well_formed<ppT>(vk)) {
            throw std::invalid_argument("verification key (write) not well-formed");
        }
        group_element_write_bytes(vk.alpha_g1, out_s);
        group_element_write_bytes(vk.beta_g2, out_s);
        group_element_write_bytes(vk.delta_g2, out_s);
        accumulation_vector_write_bytes(vk.ABC_g1, out_s);
    }

    /// groth16_snark.tcc:113-128
    static void verification_key_read_bytes(verification_key &vk, std::istream &in_s)
    {
        group_element_read_bytes(vk.alpha_g1, in_s);
        group_element_read_bytes(vk.beta_g2, in_s);
        group_element_read_bytes(vk.delta_g2, in_s);
        accumulation_vector_read_bytes(vk.ABC_g1, in_s);
        if (!is_well_formed<ppT>(vk)) {
            throw std::invalid_argument("verification key (read) not well-formed");
        }
    }

    /// groth16_snark.tcc:58-74
    static void verification_key_write_json(const verification_key &vk, std::ostream &out_s)
    {
        const size_t abc_length = vk.ABC_g1.rest.indices.size() + 1;
        out_s << "{\n  \"alpha\": " << group_element_to_json(vk.alpha_g1) << ",\n  \"beta\": " << group_element_to_json(vk.beta_g2)
              << ",\n  \"delta\": " << group_element_to_json(vk.delta_g2) << ",\n  \"ABC\": [\n    " << group_element_to_json(vk.ABC_g1.first);
        for (size_t i = 1; i < abc_length; ++i) {
            out_s << ",\n    " << group_element_to_json(vk.ABC_g1.rest.values[i - 1]);
        }
        out_s << "\n  ]\n}";
    }

    /// groth16_snark.tcc:177-191: the proving key, then the verification key -- the file prover_server keeps as
    /// ~/zeth_setup/keypair.bin (prover_server.cpp:456-463, 487-490)
    static void keypair_write_bytes(const keypair &keypair, std::ostream &out_s)
    {
        proving_key_write_bytes(keypair.pk, out_s);
        verification_key_write_bytes(keypair.vk, out_s);
    }

    static void keypair_read_bytes(keypair &keypair, std::istream &in_s)
    {
        proving_key_read_bytes(keypair.pk, in_s);
        verification_key_read_bytes(keypair.vk, in_s);
    }

    static void proof_write_bytes(const proof &proof, std::ostream &out_s)
    {
        group_element_write_bytes(proof.g_A, out_s);
        group_element_write_bytes(proof.g_B, out_s);
        group_element_write_bytes(proof.g_C, out_s);
    }

    static void proof_read_bytes(proof &proof, std::istream &in_s)
    {
        group_element_read_bytes(proof.g_A, in_s);
        group_element_read_bytes(proof.g_B, in_s);
        group_element_read_bytes(proof.g_C, in_s);
    }

    static void proof_write_json(const proof &proof, std::ostream &out_s)
    {
        out_s << "{\n  \"a\": " << group_element_to_json(proof.g_A) << ",\n  \"b\": " << group_element_to_json(proof.g_B)
              << ",\n  \"c\": " << group_element_to_json(proof.g_C) << "\n}";
    }
};

// ---- libzeth::multi_exp (libzeth/core/multi_exp.hpp:13-22) ----

/// Base points kept in HBM across calls (the `gs` of multi_exp uploaded once).
template<typename ppT, typename GroupT> class resident_bases
{
public:
    explicit resident_bases(const std::vector<GroupT> &gs, int device = 0)
        : dcp_(device_context::get_shared(device)), dc_(*dcp_), handle_(nullptr), size_(gs.size())
    {
        std::lock_guard<std::mutex> lock(dc_.mutex());
        check(zg_bases_create(dc_.ctx(), ppT::curve_id, GroupT::group_id, gs.data(), gs.size(), ZG_FORM_PLAIN, ZG_MEM_HOST, &handle_));
    }
    ~resident_bases()
    {
        std::lock_guard<std::mutex> lock(dc_.mutex());
        zg_bases_free(handle_);
    }
    resident_bases(const resident_bases &) = delete;
    resident_bases &operator=(const resident_bases &) = delete;
    size_t size() const { return size_; }

    /// Fixed-base table (zg_bases_precompute): worth its memory (12-16 times the bases) from about 2^21 bases, which
    /// are multiplied many times -- the queries of a large proving key.  Results of multi_exp do not change.
    void precompute(int window_bits = 0)
    {
        std::lock_guard<std::mutex> lock(dc_.mutex());
        check(zg_bases_precompute(handle_, window_bits));
    }

    /// sum_i fs[i] * gs[first + i]
    GroupT multi_exp(const Fr<ppT> *fs, size_t n, size_t first = 0) const
    {
        GroupT out;
        std::lock_guard<std::mutex> lock(dc_.mutex());
        check(zg_msm(handle_, first, fs, n, ZG_FORM_PLAIN, ZG_MEM_HOST, ZG_FORM_PLAIN, &out));
        return out;
    }
    GroupT multi_exp(const Fr_vector<ppT> &fs) const { return multi_exp(fs.data(), fs.size(), 0); }

private:
    std::shared_ptr<device_context> dcp_;
    device_context &dc_;
    zg_bases *handle_;
    size_t size_;
};

/// libzeth::multi_exp over several devices: the bases cut into one contiguous range per device and kept resident
/// there (zg_msm_sharded); multi_exp(fs) cuts the scalars the same way.  The "sum over chunks" of libff::multi_exp with
/// one chunk per GPU.
template<typename ppT, typename GroupT> class sharded_bases
{
public:
    sharded_bases(const std::vector<GroupT> &gs, const std::vector<int> &devices) : size_(gs.size())
    {
        if (devices.empty()) {
            throw std::invalid_argument("sharded_bases: no devices");
        }
        const size_t n = gs.size(), k = devices.size();
        try {
            for (size_t i = 0; i < k; i++) {
                const size_t lo = n * i / k, hi = n * (i + 1) / k;
                zg_ctx *c = nullptr;
                check(zg_ctx_create(devices[i], &c)); // a context of its own per shard: shards run concurrently
                ctxs_.push_back(c);
                zg_bases *b = nullptr;
                check(zg_bases_create(c, ppT::curve_id, GroupT::group_id, gs.data() + lo, hi - lo, ZG_FORM_PLAIN, ZG_MEM_HOST, &b));
                shards_.push_back(b);
                first_.push_back(lo);
            }
        } catch (...) {
            release(); // a constructor that throws runs no destructor: give back what the earlier shards hold
            throw;
        }
        first_.push_back(n);
    }
    ~sharded_bases() { release(); }
    sharded_bases(const sharded_bases &) = delete;
    sharded_bases &operator=(const sharded_bases &) = delete;
    void precompute(int window_bits = 0)
    {
        for (zg_bases *b : shards_) {
            check(zg_bases_precompute(b, window_bits));
        }
    }
    GroupT multi_exp(const Fr_vector<ppT> &fs)
    {
        if (fs.size() > size_) {
            throw std::invalid_argument("multi_exp: fewer bases than scalars"); // multi_exp.tcc:28-29
        }
        std::vector<const void *> ptrs;
        std::vector<size_t> counts;
        for (size_t i = 0; i < shards_.size(); i++) {
            const size_t lo = std::min(first_[i], fs.size()), hi = std::min(first_[i + 1], fs.size());
            ptrs.push_back(fs.data() + lo);
            counts.push_back(hi - lo);
        }
        GroupT out;
        std::lock_guard<std::mutex> lock(m_);
        check(zg_msm_sharded(shards_.data(), shards_.size(), ptrs.data(), counts.data(), ZG_FORM_PLAIN, ZG_MEM_HOST, ZG_FORM_PLAIN, &out));
        return out;
    }

private:
    void release()
    {
        for (zg_bases *b : shards_) {
            zg_bases_free(b);
        }
        for (zg_ctx *c : ctxs_) {
            zg_ctx_destroy(c);
        }
        shards_.clear();
        ctxs_.clear();
    }
    std::vector<zg_ctx *> ctxs_;
    std::vector<zg_bases *> shards_;
    std::vector<size_t> first_;
    size_t size_;
    std::mutex m_;
};

template<typename ppT, typename GroupT>
GroupT multi_exp(
    typename std::vector<GroupT>::const_iterator gs_start,
    typename std::vector<GroupT>::const_iterator gs_end,
    typename std::vector<Fr<ppT>>::const_iterator fs_start,
    typename std::vector<Fr<ppT>>::const_iterator fs_end)
{
    const size_t ng = (size_t)(gs_end - gs_start), nf = (size_t)(fs_end - fs_start);
    if (ng != nf) {
        // libff::multi_exp_filter_one_zero asserts equal lengths
        throw std::invalid_argument("multi_exp: base and scalar ranges differ in length");
    }
    if (nf == 0) {
        return GroupT::zero();
    }
    device_context &dc = device_context::get();
    zg_bases *b = nullptr;
    GroupT out;
    std::lock_guard<std::mutex> lock(dc.mutex());
    check(zg_bases_create(dc.ctx(), ppT::curve_id, GroupT::group_id, &*gs_start, ng, ZG_FORM_PLAIN, ZG_MEM_HOST, &b));
    const int s = zg_msm(b, 0, &*fs_start, nf, ZG_FORM_PLAIN, ZG_MEM_HOST, ZG_FORM_PLAIN, &out);
    zg_bases_free(b);
    check(s);
    return out;
}

template<typename ppT, typename GroupT> GroupT multi_exp(const std::vector<GroupT> &gs, const Fr_vector<ppT> &fs)
{
    // the reference asserts gs.size() >= fs.size() and gs.size() > 0 (multi_exp.tcc:28-29) and uses the first
    // fs.size() bases
    if (gs.size() < fs.size() || gs.empty()) {
        throw std::invalid_argument("multi_exp: fewer bases than scalars");
    }
    return multi_exp<ppT, GroupT>(gs.begin(), gs.begin() + fs.size(), fs.begin(), fs.end());
}

// ---- libfqfft::basic_radix2_domain<Fr> as Zeth uses it ----
template<typename ppT> class basic_radix2_domain
{
public:
    const size_t m;

    explicit basic_radix2_domain(size_t m_) : m(m_), log_m_(0)
    {
        // libfqfft throws DomainSizeException for sizes that are not a power of two or exceed the 2-adicity
        if (m_ <= 1 || (m_ & (m_ - 1)) != 0) {
            throw std::invalid_argument("basic_radix2(): expected m == (1 << log2(m)), m > 1");
        }
        while ((size_t(1) << log_m_) < m_) {
            log_m_++;
        }
        if (log_m_ > ppT::two_adicity) {
            throw std::invalid_argument("basic_radix2(): expected log2(m) <= Fr::s");
        }
    }
    void FFT(std::vector<Fr<ppT>> &a) const { run(a, 0, 0); }
    void iFFT(std::vector<Fr<ppT>> &a) const { run(a, 1, 0); }
    /// g must be Fr::multiplicative_generator, the only shift the prover and Zeth use
    void cosetFFT(std::vector<Fr<ppT>> &a, const Fr<ppT> &g) const
    {
        require_generator(g);
        run(a, 0, 1);
    }
    void icosetFFT(std::vector<Fr<ppT>> &a, const Fr<ppT> &g) const
    {
        require_generator(g);
        run(a, 1, 1);
    }

private:
    unsigned log_m_;
    static void require_generator(const Fr<ppT> &g)
    {
        if (g != Fr<ppT>(ppT::multiplicative_generator)) {
            throw std::invalid_argument("coset transforms support the shift Fr::multiplicative_generator only");
        }
    }
    void run(std::vector<Fr<ppT>> &a, int inverse, int coset) const
    {
        if (a.size() != m) {
            throw std::invalid_argument("basic_radix2: expected a.size() == this->m"); // libfqfft DomainSizeException
        }
        device_context &dc = device_context::get();
        std::lock_guard<std::mutex> lock(dc.mutex());
        check(zg_ntt(dc.ctx(), ppT::curve_id, a.data(), log_m_, inverse, coset, ZG_FORM_PLAIN, ZG_MEM_HOST));
    }
};

/// libzeth::evaluator_from_lagrange (evaluator_from_lagrange.tcc:14-45): scatter the Lagrange factors into a
/// domain-sized vector, iFFT to coefficients, multi_exp against the powers.
template<typename ppT, typename GroupT> class evaluator_from_lagrange
{
public:
    evaluator_from_lagrange(const std::vector<GroupT> &powers, const basic_radix2_domain<ppT> &domain) : powers_(powers), domain_(domain)
    {
        if (powers.size() < domain.m) {
            throw std::invalid_argument("evaluator_from_lagrange: fewer powers than the domain size");
        }
    }
    GroupT evaluate_from_lagrange_factors(const std::map<size_t, Fr<ppT>> &lagrange_factors) const
    {
        std::vector<Fr<ppT>> coefficients(domain_.m, Fr<ppT>::zero());
        for (const auto &it : lagrange_factors) {
            if (it.first >= domain_.m) {
                throw std::invalid_argument("evaluator_from_lagrange: factor index outside the domain");
            }
            coefficients[it.first] = it.second;
        }
        domain_.iFFT(coefficients);
        return multi_exp<ppT, GroupT>(powers_, coefficients);
    }

private:
    const std::vector<GroupT> &powers_;
    const basic_radix2_domain<ppT> &domain_;
};

// ---- ceremony kernels (libzeth/mpc/groth16) ----

/// compute_lagrange_from_powers (powersoftau_utils.tcc:125-134): [x^i] -> [L_j(x)] in place, i.e.
/// libfqfft::_basic_radix2_FFT<Fr, Gr>(powers, omega_inv) followed by the scaling by n^-1; the body of
/// powersoftau_compute_lagrange_evaluations (:830-902), which rejects sizes that are not a power of two.
template<typename ppT, typename GroupT> void compute_lagrange_from_powers(std::vector<GroupT> &powers)
{
    const size_t n = powers.size();
    if (n == 0 || (n & (n - 1)) != 0) {
        throw std::invalid_argument("non-pow-2 domain");
    }
    unsigned log_n = 0;
    while ((size_t(1) << log_n) < n) {
        log_n++;
    }
    device_context &dc = device_context::get();
    std::lock_guard<std::mutex> lock(dc.mutex());
    check(zg_group_fft(dc.ctx(), ppT::curve_id, GroupT::group_id, powers.data(), log_n, 1, ZG_FORM_PLAIN));
}

/// a <- k * a for every element: the delta_j^-1 update of L_g1 / H_g1 (phase2.tcc:447-470)
template<typename ppT, typename GroupT> void batch_scalar_mul(std::vector<GroupT> &points, const Fr<ppT> &k)
{
    device_context &dc = device_context::get();
    std::lock_guard<std::mutex> lock(dc.mutex());
    check(zg_group_scale(dc.ctx(), ppT::curve_id, GroupT::group_id, points.data(), points.size(), k.data, ZG_FORM_PLAIN));
}

} // namespace libzeth_gpu

#endif // LIBZETH_GPU_GROTH16_SNARK_HPP
