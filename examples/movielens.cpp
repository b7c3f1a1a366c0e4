// The crate's README example (src/lib.rs:22-58) against the C++ host layer of the MI355X engine.
//
//   python -m sbr_rs_amd.build                      # builds sbr_rs_amd/libsbr_hip.so
//   g++ -std=c++17 -O2 -Iinclude examples/movielens.cpp -o movielens -Lsbr_rs_amd -lsbr_hip -Wl,-rpath,$PWD/sbr_rs_amd
//   ./movielens data.csv                             # user_id,item_id,rating,timestamp
#include <chrono>
#include <cstdio>

#include "sbr.hpp"

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <movielens data.csv> [num_threads] [partition_item_table 0|1]\n", argv[0]);
        return 2;
    }
    auto data = sbr::datasets::download_movielens_100k(argv[1]);
    std::array<std::uint8_t, 16> seed;
    seed.fill(42);
    auto rng = sbr::XorShiftRng::from_seed(seed);
    auto [train, test] = sbr::data::user_based_split(data, rng, 0.2f);
    const auto train_mat = train.to_compressed(), test_mat = test.to_compressed();
    std::printf("Train: %zu, test: %zu\n", train.len(), test.len());

    auto model = sbr::models::lstm::Hyperparameters::new_(data.num_items(), 32)
                     .embedding_dim(32)
                     .learning_rate(0.16f)
                     .l2_penalty(0.0004f)
                     .lstm_variant(sbr::models::lstm::LSTMVariant::Normal)
                     .loss(sbr::models::Loss::WARP)
                     .optimizer(sbr::models::Optimizer::Adagrad)
                     .num_epochs(10)
                     .num_threads(argc > 2 ? std::atoi(argv[2]) : 1)          // device replicas in this process
                     .partition_item_table(argc > 3 && std::atoi(argv[3]))    // store the item table once across them
                     .rng(rng)
                     .build();

    const auto start = std::chrono::steady_clock::now();
    const float loss = model.fit(train_mat).unwrap();
    const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
    const float train_mrr = sbr::evaluation::mrr_score(model, train_mat).unwrap();
    const float test_mrr = sbr::evaluation::mrr_score(model, test_mat).unwrap();
    std::printf("Train MRR %g at loss %g and test MRR %g (in %.3f s)\n", train_mrr, loss, test_mrr, elapsed);

    // online use: a user's representation from their history, scores for a few items
    const auto user = model.user_representation({50, 181, 258}).unwrap();
    const auto scores = model.predict(user, {1, 100, 300}).unwrap();
    std::printf("scores: %g %g %g\n", scores[0], scores[1], scores[2]);
    return 0;
}
