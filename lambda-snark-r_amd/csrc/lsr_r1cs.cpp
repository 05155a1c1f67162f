// SEAL/NTL-free R1CS shim (host only; see include/lambda_snark/r1cs.h for scope and citations).
#include <cstdint>
#include <new>
#include <vector>

#include "lambda_snark/r1cs.h"

namespace {

using i128 = __int128;

// NTL::conv(ZZ_p, static_cast<long>(x)) — the 64-bit pattern read as a signed integer, reduced into [0, q)
uint64_t signed_residue(uint64_t x, uint64_t q) {
    const i128 r = static_cast<i128>(static_cast<int64_t>(x)) % static_cast<i128>(q);
    return static_cast<uint64_t>(r < 0 ? r + static_cast<i128>(q) : r);
}

struct System {
    std::vector<SparseEntry> a, b, c;
    uint32_t rows = 0, cols = 0;
    uint64_t q = 0;
};

struct OutOfRange {};

std::vector<uint64_t> sparse_mv(const std::vector<SparseEntry>& m, uint32_t rows, const uint64_t* z, size_t len, uint64_t q) {
    std::vector<uint64_t> out(rows, 0);
    for (const SparseEntry& e : m) {
        if (e.col >= len || e.row >= rows) throw OutOfRange{};
        const unsigned __int128 prod = static_cast<unsigned __int128>(signed_residue(e.value, q)) * signed_residue(z[e.col], q);
        out[e.row] = static_cast<uint64_t>((prod + signed_residue(out[e.row], q)) % q);
    }
    return out;
}

}  // namespace

extern "C" {

LambdaSnarkError lambda_snark_r1cs_create(const SparseMatrix* A, const SparseMatrix* B, const SparseMatrix* C, uint64_t modulus,
                                          void** out_r1cs) noexcept {
    if (!A || !B || !C || !out_r1cs) return LAMBDA_SNARK_ERR_NULL_PTR;
    if (A->n_rows != B->n_rows || B->n_rows != C->n_rows || A->n_cols != B->n_cols || B->n_cols != C->n_cols) return LAMBDA_SNARK_ERR_INVALID_PARAMS;
    if (modulus < 2) return LAMBDA_SNARK_ERR_INVALID_PARAMS;   // NTL::ZZ_p::init rejects it
    try {
        auto* s = new System;
        s->rows = A->n_rows;
        s->cols = A->n_cols;
        s->q = modulus;
        const SparseMatrix* src[3] = {A, B, C};
        std::vector<SparseEntry>* dst[3] = {&s->a, &s->b, &s->c};
        for (int i = 0; i < 3; ++i)
            if (src[i]->n_entries) {
                if (!src[i]->entries) { delete s; return LAMBDA_SNARK_ERR_NULL_PTR; }
                dst[i]->assign(src[i]->entries, src[i]->entries + src[i]->n_entries);
            }
        *out_r1cs = s;
        return LAMBDA_SNARK_OK;
    } catch (const std::bad_alloc&) {
        return LAMBDA_SNARK_ERR_ALLOC_FAILED;
    } catch (...) {
        return LAMBDA_SNARK_ERR_CRYPTO_FAILED;
    }
}

LambdaSnarkError lambda_snark_r1cs_validate_witness(void* r1cs, const R1CSWitness* witness, bool* out_valid) noexcept {
    if (!r1cs || !witness || !out_valid) return LAMBDA_SNARK_ERR_NULL_PTR;
    const auto* s = static_cast<const System*>(r1cs);
    if (witness->len != s->cols || !witness->values || witness->values[0] != 1) return LAMBDA_SNARK_ERR_INVALID_PARAMS;   // r1cs.cpp:96-105
    try {
        const auto az = sparse_mv(s->a, s->rows, witness->values, witness->len, s->q);
        const auto bz = sparse_mv(s->b, s->rows, witness->values, witness->len, s->q);
        const auto cz = sparse_mv(s->c, s->rows, witness->values, witness->len, s->q);
        bool ok = true;
        for (uint32_t i = 0; i < s->rows && ok; ++i)
            ok = static_cast<uint64_t>(static_cast<unsigned __int128>(az[i]) * bz[i] % s->q) == cz[i];
        *out_valid = ok;
        return LAMBDA_SNARK_OK;
    } catch (const OutOfRange&) {
        return LAMBDA_SNARK_ERR_CRYPTO_FAILED;   // std::out_of_range falls into ffi.cpp's catch (...)
    } catch (...) {
        return LAMBDA_SNARK_ERR_CRYPTO_FAILED;
    }
}

void lambda_snark_r1cs_free(void* r1cs) noexcept { delete static_cast<System*>(r1cs); }
uint32_t lambda_snark_r1cs_num_constraints(void* r1cs) noexcept { return r1cs ? static_cast<System*>(r1cs)->rows : 0; }
uint32_t lambda_snark_r1cs_num_variables(void* r1cs) noexcept { return r1cs ? static_cast<System*>(r1cs)->cols : 0; }

}  // extern "C"
