// shim_keys.hpp -- resident evaluation keys: one device copy per host key, verified by a keyed fingerprint of every word.
// Part of the MPI-typed surface: one translation unit (mpi_shim.hip includes these fragments in order); split by concern in round 4.
#pragma once

namespace {

// Evaluation keys are 2 x dim x n words (47 MB at the headline shape) and the same key multiplies many ciphertexts: the device
// copy is kept, identified by the caller's two pointers and a fingerprint of EVERY word, limb by limb (the reference reads the key
// it is given on every call, src/he-mult.c:60-64: a key edited in place, in however few words, must multiply as edited).  A call at a
// lower level reads fewer limbs of the same key (dimB shrinks with q_l, :51): the resident copy serves every prefix of itself, checked
// over just the limbs in use -- one copy per key, not one per level.  For a key that is resident the fingerprint is computed by the host
// threads WHILE the device works with the resident copy (they would wait for it otherwise: resident_key / key_still_valid; a mismatch
// repeats the device work with the fresh key); for a new key, next to the ciphertext conversions.  gpq_mpi_shim_set_key_check(0) goes
// back to ~1000 sampled words (of the exact length in use) for callers that never edit a key in place.
struct KeySlot { const uint64_t *h0, *h1; unsigned limbs, n; bool full; std::vector<uint64_t> print; void *d0, *d1; size_t cap_words; uint64_t used; };
std::vector<KeySlot> g_keys;
uint64_t g_key_clock = 0;
size_t g_key_slots = 64;      // resident keys (rlk, ck, the rotation keys in use): 45 MiB each at the headline shape (2.9 GB of the 288 GB when all are in use); gpq_mpi_shim_set_key_slots
bool g_key_check_full = true; // gpq_mpi_shim_set_key_check
uint64_t key_print_sampled(const uint64_t *a, const uint64_t *b, size_t words) {
  const HashKey &key = hash_key();
  uint64_t h = key.init[3];
  auto mix = [&](uint64_t v) { h = (h ^ v) * key.fold; h ^= h >> 29; };
  const size_t step = words > 512 ? words / 509 : 1;     // ~512 samples of each polynomial, plus both ends
  for (size_t i = 0; i < words; i += step) { mix(a[i]); mix(b[i]); }
  for (size_t i = 0; i < 8 && i < words; ++i) { mix(a[i]); mix(b[i]); mix(a[words - 1 - i]); mix(b[words - 1 - i]); }
  return h;
}
struct KeyPrint {                 // fingerprint of the first `limbs` limbs of a host key: one piece per limb, any thread may compute any piece
  const uint64_t *h0, *h1; unsigned limbs, n; bool full; unsigned parts; std::vector<uint64_t> part;
  std::function<void(unsigned)> task;
  KeyPrint(const he_evk_t *key, unsigned limbs_, unsigned n_) : h0(key->p0.coeffs), h1(key->p1.coeffs), limbs(limbs_), n(n_), full(g_key_check_full) {
    parts = full ? limbs : 1;
    part.assign(parts, 0);
    task = [this](unsigned t) {
      if (!full) { part[t] = key_print_sampled(h0, h1, (size_t)limbs * n); return; }
      const size_t lo = (size_t)t * n, hi = lo + n;
      part[t] = hash_words(h0, lo, hi) * hash_key().fold + hash_words(h1, lo, hi);
    };
  }
  size_t words() const { return (size_t)limbs * n; }
  bool covered_by(const KeySlot &k) const {                 // the slot's shape can serve this request at all
    return k.h0 == h0 && k.h1 == h1 && k.n == n && k.full == full && (full ? k.limbs >= limbs : k.limbs == limbs);
  }
  bool matches(const KeySlot &k) const {                    // ... and (part[] computed) holds the words the caller holds now
    return covered_by(k) && std::equal(part.begin(), part.end(), k.print.begin());
  }
};
void drop_key_slot(size_t i) {
  (void)gpq_stream_sync(nullptr);
  (void)gpq_free(g_keys[i].d0); (void)gpq_free(g_keys[i].d1);
  g_keys.erase(g_keys.begin() + i);
}
void forget_key_at(const uint64_t *h0, const uint64_t *h1) {          // the host key at these addresses was rewritten (he_gen*k)
  for (size_t i = g_keys.size(); i-- > 0;)
    if (g_keys[i].h0 == h0 || g_keys[i].h1 == h1 || g_keys[i].h0 == h1 || g_keys[i].h1 == h0) drop_key_slot(i);
}
// A resident copy of the host key at these addresses that covers the limbs in use, whatever its fingerprint: he_mul / he_rot / he_conj
// start the device work with it at once and verify the fingerprint on the host threads WHILE the device works (they would otherwise
// wait for it); a mismatch -- the key was edited in place since -- uploads the key and runs the device work again.
KeySlot *resident_key(const KeyPrint &kp) {
  for (KeySlot &k : g_keys)
    if (kp.covered_by(k)) { k.used = ++g_key_clock; return &k; }
  return nullptr;
}
bool key_still_valid(KeySlot *slot, KeyPrint &kp) {
  if (kp.parts < 2) kp.task(0); else workers().run(kp.parts, kp.task);
  return kp.matches(*slot);
}

void key_on_device(const KeyPrint &kp, uint64_t **d0, uint64_t **d1) {
  const uint64_t *h0 = kp.h0, *h1 = kp.h1;
  const size_t words = kp.words();
  for (KeySlot &k : g_keys)
    if (kp.matches(k)) { k.used = ++g_key_clock; *d0 = (uint64_t *)k.d0; *d1 = (uint64_t *)k.d1; return; }
  KeySlot slot{h0, h1, kp.limbs, kp.n, kp.full, kp.part, nullptr, nullptr, words, ++g_key_clock};
  size_t victim = g_keys.size();
  for (size_t i = 0; i < g_keys.size(); ++i)               // same host key in another state / at another length, else the least recently used
    if (g_keys[i].h0 == h0 && g_keys[i].h1 == h1) victim = i;
  if (victim == g_keys.size() && g_keys.size() >= g_key_slots) {
    victim = 0;
    for (size_t i = 1; i < g_keys.size(); ++i) if (g_keys[i].used < g_keys[victim].used) victim = i;
  }
  if (victim < g_keys.size()) {
    if (g_keys[victim].cap_words >= words) { slot.d0 = g_keys[victim].d0; slot.d1 = g_keys[victim].d1; slot.cap_words = g_keys[victim].cap_words; g_keys.erase(g_keys.begin() + victim); }
    else drop_key_slot(victim);
  }
  if (!slot.d0 && (gpq_malloc(&slot.d0, words * 8) != GPQ_OK || gpq_malloc(&slot.d1, words * 8) != GPQ_OK)) die("device allocation failed");
  if (gpq_upload(slot.d0, h0, words * 8, nullptr) != GPQ_OK || gpq_upload(slot.d1, h1, words * 8, nullptr) != GPQ_OK) die("upload failed");
  g_keys.push_back(slot);
  *d0 = (uint64_t *)slot.d0; *d1 = (uint64_t *)slot.d1;
}

}  // namespace
