// pipeline.cpp -- the optimiser run: config -> clouds -> plane -> discrete search
// -> gradient descent -> results.txt / new_experiments_errors.txt.
//
// Host-side mirror of src/camera_calibration_optimization.cpp: main (:560-581),
// Optimizer::getInputs (:148-278), discreteCombinatorialOptimization (:526-558),
// gradientDescent (:411-465), gradientDescentWithPlane (:466-524) and
// Optimizer::printError (:348-407).  Every objective value, the plane fit and
// both searches go through the C ABI of include/ccal.h; this file only parses,
// uploads, and writes the two text files in the reference's format (default
// ostream formatting, append mode).
#include "../../../include/ccal.h"
#include "cloud_io.hpp"
#include "xml_lite.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>
#include <iostream>
#include <iterator>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

using namespace ccal_host;

namespace {

// A failure with the status the C ABI reports for it.
struct Failure : std::runtime_error {
    int status;
    Failure(int s, const std::string& what) : std::runtime_error(what), status(s) {}
};

struct Inputs {
    std::vector<Cloud> raw, down;
    std::vector<double> base_T_flange;   // per cloud, already mapped (:290)
    double pose_init[6];
    ccal_grid grid;
    int n_gpus = 1;
    bool share_gpus = false;  // <device><share>: more contexts than GPUs, round-robin (rehearsals, tests)
    int search_mode = CCAL_SEARCH_SEPARABLE;   // same result as the dense sweep, O(rotations x points) work
    ccal_gd_opts gd;
    std::string config_out;   // optional <output><config>: refined config written here
};

std::string slurp(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open config " + path);
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

double num(const XmlNode& root, const std::string& path, double dflt)
{
    const XmlNode* n = root.find(path);
    if (!n) return dflt;
    return std::strtod(trim(n->text).c_str(), nullptr);
}

// Optimizer::getInputs (:148-237) without the plane fit
Inputs load_inputs(const std::string& config_file, bool chatty)
{
    Inputs in;
    const std::string xml = slurp(config_file);
    auto root = XmlParser(xml).parse();
    if (chatty) std::cout << "Config File: " << config_file << std::endl;
    const std::string camera_metric = trim(root->get("data.camera.metric", "m"));   // :156
    const XmlNode* clouds = root->find("data.camera.clouds");
    if (!clouds || clouds->children.empty()) throw std::runtime_error("config: no data.camera.clouds entries");

    std::vector<int> mapping;
    std::vector<std::string> filenames;
    for (const auto& c : clouds->children) {                                       // :158
        const std::string filename = trim(c->text);
        const int cloud_id = file_id(filename);                                    // :161
        if (chatty) std::cout << "Cloud Number: " << cloud_id << std::endl;
        mapping.push_back(cloud_id - 1);                                           // :163
        filenames.push_back(filename);
    }
    // read (:165) and drop z > 0.8 / NaN (:167-169); the files are independent, so a few host
    // threads take them in turn.  The first failing file in document order is the one reported.
    // The voxel grid of :199-203 runs on the device after the upload.
    in.raw.resize(filenames.size());
    {
        std::vector<std::exception_ptr> errors(filenames.size());
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (size_t k = next++; k < filenames.size(); k = next++) {
                try {
                    Cloud kept = filter_z(read_point_cloud(filenames[k], camera_metric), 0.8);
                    if (kept.empty()) throw std::runtime_error(filenames[k] + ": no points with z <= 0.8");
                    in.raw[k] = std::move(kept);
                } catch (...) {
                    errors[k] = std::current_exception();
                }
            }
        };
        const size_t n_threads = std::min<size_t>({filenames.size(), 8, std::max(1u, std::thread::hardware_concurrency())});
        std::vector<std::thread> pool;
        for (size_t t = 1; t < n_threads; ++t) pool.emplace_back(work);
        work();
        for (auto& t : pool) t.join();
        for (auto& e : errors)
            if (e) std::rethrow_exception(e);
    }

    const XmlNode* ik = root->find("data.camera.transformations.inverse_kinematics.location");
    if (!ik) throw std::runtime_error("config: inverse_kinematics.location missing");  // ptree::get throws
    const std::string ik_file = trim(ik->text);
    const std::string ik_metric = trim(root->get("data.camera.transformations.inverse_kinematics.metric", "m"));
    auto iks = read_transformations(ik_file, ik_metric);                           // :210
    in.base_T_flange.resize(16 * mapping.size());
    for (size_t k = 0; k < mapping.size(); ++k) {
        if (mapping[k] < 0 || mapping[k] >= (int)iks.size())
            throw std::runtime_error("cloud id " + std::to_string(mapping[k] + 1) + " has no row in " + ik_file);
        ccal_pose_to_matrix(iks[mapping[k]].data(), &in.base_T_flange[16 * k]);
    }

    // getTransVector (:70-98)
    const std::string s = "data.camera.transformations.approximate_transformation";
    const std::string angle = trim(root->get(s + ".angle", "radian"));
    const std::string metric = trim(root->get(s + ".metric", "m"));
    const std::string value = root->get(s + ".value", "0,0,0,0,0,0");
    for (double& v : in.pose_init) v = 0.0;
    parse_pose_csv(value, metric_scale(metric), angle == "degree", in.pose_init);
    if (chatty) {
        std::cout << value << std::endl;
        for (double v : in.pose_init) std::cout << v << " ";
        std::cout << std::endl;
    }

    // optional extensions (absent -> the reference's fixed 11^6 grid, one GPU)
    ccal_grid_default(&in.grid);
    for (int k = 0; k < 6; ++k) {
        const std::string grp = k < 3 ? "data.search.translation_mm" : "data.search.rotation_deg";
        in.grid.min[k] = num(*root, grp + ".min", in.grid.min[k]);
        in.grid.step[k] = num(*root, grp + ".step", in.grid.step[k]);
        in.grid.count[k] = (int)num(*root, grp + ".count", in.grid.count[k]);
    }
    in.n_gpus = (int)num(*root, "data.device.gpus", 1);
    in.share_gpus = num(*root, "data.device.share", 0) != 0;
    {
        const std::string mode = trim(root->get("data.search.mode", ""));
        if (mode == "dense") in.search_mode = CCAL_SEARCH_DENSE;
        else if (mode == "separable" || mode.empty()) in.search_mode = CCAL_SEARCH_SEPARABLE;
        else throw Failure(CCAL_ERR_ARG, "config: <search><mode> must be dense or separable");
    }
    ccal_gd_default_opts(&in.gd);
    in.gd.max_iterations = (int)num(*root, "data.search.gd_max_iterations", in.gd.max_iterations);
    in.config_out = trim(root->get("data.output.config", ""));
    return in;
}

struct Check {
    ccal_ctx* ctx;
    void operator()(int rc, const char* what) const
    {
        if (rc != CCAL_OK)
            throw Failure(rc, std::string(what) + ": " + (ctx ? ccal_last_error(ctx) : ccal_status_string(rc)));
    }
};

// One process per GPU: the run's place among its peers, from the environment of the launcher.
//   CCAL_WORLD_SIZE, CCAL_RANK   number of processes and this one's index (default 1, 0)
//   CCAL_LOCAL_RANK              GPU of this process (default: the rank)
//   CCAL_COMM_FILE               rendezvous file: rank 0 writes the 128-byte communicator id there
//                                (and removes it once every rank has joined), the others wait for it.
//                                When set, the exchange runs even with one process.
//   CCAL_COMM_NONCE              any string naming THIS job (the same on every rank).  It is stored
//                                behind the id, and a rank accepts only a file that carries its own
//                                nonce: a file left behind by a crashed earlier job is then never
//                                mistaken for this job's (its id would block ncclCommInitRank for good).
//   CCAL_COMM_TIMEOUT_S          seconds a rank waits for the file, inside ccal_comm_create and inside
//                                the exchange before it gives up and ends the process (default 600).
// RCCL collectives cannot be cancelled: a process whose peers never arrive can only be ended.  A
// launcher should still end the whole job when one process dies; the deadline is the backstop.
struct Peers {
    int world = 1, rank = 0, local_rank = 0;
    int timeout_s = 600;
    std::string comm_file, nonce;
    bool exchange() const { return world > 1 || !comm_file.empty(); }
};

Peers peers_from_env()
{
    Peers p;
    auto geti = [](const char* name, int dflt) {
        const char* v = std::getenv(name);
        return v && *v ? std::atoi(v) : dflt;
    };
    p.world = geti("CCAL_WORLD_SIZE", 1);
    p.rank = geti("CCAL_RANK", 0);
    p.local_rank = geti("CCAL_LOCAL_RANK", p.rank);
    p.timeout_s = std::max(1, geti("CCAL_COMM_TIMEOUT_S", 600));
    if (const char* f = std::getenv("CCAL_COMM_FILE")) p.comm_file = f;
    if (const char* f = std::getenv("CCAL_COMM_NONCE")) p.nonce = f;
    if (p.world < 1 || p.rank < 0 || p.rank >= p.world) throw Failure(CCAL_ERR_ARG, "CCAL_RANK / CCAL_WORLD_SIZE out of range");
    if (p.world > 1 && p.comm_file.empty()) throw Failure(CCAL_ERR_ARG, "CCAL_WORLD_SIZE > 1 needs CCAL_COMM_FILE");
    return p;
}

// Ends the process when a collective has not returned in time (its peers died or never came).
class Deadline {
public:
    Deadline(int seconds, const char* what)
    {
        watcher_ = std::thread([this, seconds, what]() {
            std::unique_lock<std::mutex> lk(m_);
            if (!cv_.wait_for(lk, std::chrono::seconds(seconds), [this]() { return done_; })) {
                std::fprintf(stderr, "camera_calibration_optimization: %s did not complete within %d s "
                                     "(a peer process is missing); giving up\n", what, seconds);
                std::fflush(stderr);
                std::_Exit(3);
            }
        });
    }
    ~Deadline()
    {
        { std::lock_guard<std::mutex> lk(m_); done_ = true; }
        cv_.notify_all();
        watcher_.join();
    }
private:
    std::mutex m_;
    std::condition_variable cv_;
    bool done_ = false;
    std::thread watcher_;
};

// The rendezvous file: 128 bytes of communicator id, then the job's nonce.
bool read_comm_file(const std::string& path, const std::string& nonce, char id[CCAL_COMM_ID_BYTES])
{
    std::ifstream f(path, std::ios::binary);
    if (!f || !f.read(id, CCAL_COMM_ID_BYTES) || f.gcount() != (std::streamsize)CCAL_COMM_ID_BYTES) return false;
    std::string tail((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return tail == nonce;
}

ccal_comm* join_peers(ccal_ctx* ctx, const Peers& p)
{
    char id[CCAL_COMM_ID_BYTES];
    const std::string tmp = p.comm_file + ".tmp";
    if (p.rank == 0) {
        std::remove(p.comm_file.c_str());       // whatever an earlier job left there
        if (ccal_comm_unique_id(id) != CCAL_OK) throw Failure(CCAL_ERR_DEVICE, "ccal_comm_unique_id failed (librccl missing?)");
        {
            std::ofstream f(tmp, std::ios::binary);
            f.write(id, sizeof id);
            f.write(p.nonce.data(), (std::streamsize)p.nonce.size());
            if (!f) throw Failure(CCAL_ERR_IO, "cannot write " + tmp);
        }
        if (std::rename(tmp.c_str(), p.comm_file.c_str()) != 0) throw Failure(CCAL_ERR_IO, "cannot create " + p.comm_file);
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        while (!read_comm_file(p.comm_file, p.nonce, id)) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(p.timeout_s))
                throw Failure(CCAL_ERR_IO, "no communicator id of this job in " + p.comm_file + " after " +
                                               std::to_string(p.timeout_s) + " s");
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    }
    ccal_comm* comm = nullptr;
    {
        Deadline d(p.timeout_s, "joining the communicator");
        Check{ctx}(ccal_comm_create(ctx, p.world, p.rank, id, &comm), "comm_create");
    }
    if (p.rank == 0) std::remove(p.comm_file.c_str());   // every rank has read it: comm_create is collective
    return comm;
}

void upload(ccal_ctx* ctx, int which, const std::vector<Cloud>& clouds, const std::vector<double>& B)
{
    std::vector<const float*> ptrs;
    std::vector<size_t> sizes;
    for (const auto& c : clouds) { ptrs.push_back(c.data()); sizes.push_back(c.size() / 3); }
    Check{ctx}(ccal_upload_clouds(ctx, which, (int)clouds.size(), ptrs.data(), sizes.data(), B.data()), "upload");
}

// Optimizer::printError (:348-407)
void print_error(ccal_ctx* ctx, const double pose[6], int n_clouds, std::ostream& errorfile)
{
    std::vector<double> mean(n_clouds), mx(n_clouds);
    double cost;
    Check{ctx}(ccal_eval(ctx, CCAL_DOWNSAMPLED, 1, pose, nullptr, &cost, mean.data(), mx.data()), "eval");
    for (int j = 0; j < n_clouds; ++j)
        errorfile << "Downsampled No: " << j << " Error Avg: " << mean[j] * 1000.0
                  << " Errom Max: " << mx[j] * 1000.0 << std::endl;                  // :376
    Check{ctx}(ccal_eval(ctx, CCAL_RAW, 1, pose, nullptr, &cost, mean.data(), mx.data()), "eval");
    for (int j = 0; j < n_clouds; ++j)
        errorfile << "Real No: " << j << " Error Avg: " << mean[j] * 1000.0
                  << " Errom Max: " << mx[j] * 1000.0 << std::endl;                  // :404
    errorfile << "--------------------------------------------" << std::endl;        // :406
}

} // namespace

extern "C" int ccal_run_config(const char* config_path, const char* results_path,
                               const char* errors_path, int verbose, char* errbuf, size_t errbuf_len)
{
    std::vector<ccal_ctx*> ctxs;
    ccal_comm* comm = nullptr;
    bool exchanged = false;        // this process has taken part in the arg-min exchange
    int comm_timeout_s = 600;
    auto cleanup = [&]() {
        if (comm) ccal_comm_destroy(comm);
        comm = nullptr;
        for (auto* c : ctxs) ccal_destroy(c);
        ctxs.clear();
    };
    try {
        if (!config_path) throw Failure(CCAL_ERR_ARG, "null config path");
        const Peers peers = peers_from_env();
        comm_timeout_s = peers.timeout_s;
        const bool writer = peers.rank == 0;              // one process writes the two text files
        const bool chatty = verbose != 0 && writer;
        std::ofstream outfile, errorfile;                 // left closed on the other ranks: writes go nowhere
        if (writer) {
            outfile.open(results_path ? results_path : "results.txt", std::ios_base::app);                     // :567
            errorfile.open(errors_path ? errors_path : "new_experiments_errors.txt", std::ios_base::app);     // :569
            if (!outfile || !errorfile) throw Failure(CCAL_ERR_IO, "cannot open the result files for appending");
            outfile << "Results: " << config_path << std::endl;                                                // :571
            errorfile << "Results: " << config_path << std::endl;                                              // :572
        }

        Inputs in = load_inputs(config_path, chatty);
        const int C = (int)in.raw.size();
        const int avail = ccal_device_count();
        if (avail < 1) throw Failure(CCAL_ERR_DEVICE, "no GPU visible: the optimiser has no CPU path");
        // contexts of this process: one per GPU (one host thread each); with <device><share> the
        // configured number even on fewer GPUs, round-robin.  One process per GPU: exactly one.
        int n_ctx = std::max(1, in.share_gpus ? in.n_gpus : std::min(in.n_gpus, avail));
        if (peers.world > 1) n_ctx = 1;
        for (int d = 0; d < n_ctx; ++d) {
            ccal_ctx* c = nullptr;
            const int dev = peers.world > 1 ? peers.local_rank % avail : d % avail;
            Check{nullptr}(ccal_create(dev, &c), "ccal_create");
            ctxs.push_back(c);
        }
        ccal_ctx* ctx = ctxs[0];
        if (peers.exchange()) comm = join_peers(ctx, peers);
        // raw clouds go to the device once; VoxelGrid(0.02) (:196-205) runs there
        upload(ctx, CCAL_RAW, in.raw, in.base_T_flange);
        std::vector<size_t> n_down(C);
        Check{ctx}(ccal_downsample_raw(ctx, 0.02f, n_down.data()), "downsample_raw");
        if (chatty)
            for (int j = 0; j < C; ++j) std::cout << "Filtered Clouds Size: " << n_down[j] << std::endl;   // :204
        if (n_ctx > 1) {
            in.down.resize(C);
            for (int j = 0; j < C; ++j) {
                in.down[j].resize(3 * n_down[j]);
                Check{ctx}(ccal_download_cloud(ctx, CCAL_DOWNSAMPLED, j, in.down[j].data()), "download_cloud");
            }
            for (int d = 1; d < n_ctx; ++d) upload(ctxs[d], CCAL_DOWNSAMPLED, in.down, in.base_T_flange);
        }

        // plane from every raw point carried to the base frame by the initial pose (:240-277)
        float plane[4];
        Check{ctx}(ccal_fit_plane_lsq(ctx, CCAL_RAW, in.pose_init, plane), "fit_plane_lsq");
        for (auto* c : ctxs) Check{c}(ccal_set_plane(c, plane), "set_plane");
        for (auto* c : ctxs) Check{c}(ccal_set_search_mode(c, in.search_mode), "set_search_mode");
        if (chatty) {
            std::cout << "Plane Equation 2: " << std::endl;
            for (int i = 0; i < 4; ++i) std::cout << plane[i] << " ";
            std::cout << std::endl << "Starting optimization" << std::endl;
        }

        // ---- discreteCombinatorialOptimization (:526-558) -----------------------------
        double true_err;
        Check{ctx}(ccal_eval(ctx, CCAL_DOWNSAMPLED, 1, in.pose_init, nullptr, &true_err, nullptr, nullptr), "eval");
        if (chatty) std::cout << "True: " << true_err;                                                      // :529
        const int64_t total = ccal_grid_size(&in.grid);
        if (total <= 0) throw Failure(CCAL_ERR_ARG, "config: empty search grid");
        // this process's contiguous share of the grid, cut again over its contexts
        const int64_t p_begin = total * peers.rank / peers.world, p_end = total * (peers.rank + 1) / peers.world;
        std::vector<double> costs(n_ctx, 1e9), poses(6 * (size_t)n_ctx);
        std::vector<int64_t> idxs(n_ctx, -1);
        std::vector<int> rcs(n_ctx, CCAL_OK);
        std::vector<std::thread> th;
        for (int d = 0; d < n_ctx; ++d) {
            const int64_t b = p_begin + (p_end - p_begin) * d / n_ctx, e = p_begin + (p_end - p_begin) * (d + 1) / n_ctx;
            th.emplace_back([&, d, b, e]() {
                rcs[d] = ccal_grid_search(ctxs[d], CCAL_DOWNSAMPLED, in.pose_init, &in.grid, b, e, &idxs[d],
                                          &costs[d], &poses[6 * (size_t)d], nullptr);
            });
        }
        for (auto& t : th) t.join();
        for (int d = 0; d < n_ctx; ++d) Check{ctxs[d]}(rcs[d], "grid_search");
        const int w = ccal_argmin_merge(n_ctx, costs.data(), idxs.data());
        double best_cost = w >= 0 ? costs[w] : 1e9;
        int64_t best_index = w >= 0 ? idxs[w] : -1;
        if (comm) {
            // the one exchange between processes: (cost, index) records over RCCL, lexicographic minimum
            Deadline d(peers.timeout_s, "the arg-min exchange");
            exchanged = true;
            Check{ctx}(ccal_argmin_allgather(ctx, comm, best_cost, best_index, &best_cost, &best_index, nullptr),
                       "argmin_allgather");
        }
        double best[6];
        std::memcpy(best, in.pose_init, sizeof best);
        if (best_index >= 0) ccal_grid_pose(&in.grid, in.pose_init, best_index, best);
        if (!writer) {            // gradient descent runs on one GPU: the other processes are done
            cleanup();
            return CCAL_OK;
        }
        if (chatty) {
            std::cout << "Error: " << best_cost << std::endl << "Best Transformation:" << std::endl;
            for (double v : best) std::cout << v << " ";
            std::cout << std::endl;
        }
        outfile << "Flange Transformation From Discrete Optimization" << std::endl;                         // :553
        for (int i = 0; i < 5; ++i) outfile << best[i] << ", ";
        outfile << best[5] << std::endl;

        // ---- gradientDescent (:411-465) -----------------------------------------------
        ccal_gd_result r6;
        Check{ctx}(ccal_gd_minimize(ctx, CCAL_DOWNSAMPLED, 6, best, &in.gd, &r6), "gd_minimize");
        if (chatty)
            std::cout << "Done! Converged: " << (r6.converged ? "true" : "false") << " Iterations: " << r6.iterations
                      << std::endl << "Final fval: " << r6.fval << std::endl;
        outfile << "Gradient Descent on Flange Transformation..." << std::endl;
        outfile << "Iterations: " << r6.iterations << " Converged: " << (r6.converged ? "true" : "false")
                << " Final fval: " << r6.fval << std::endl;
        outfile << "Flange Transformation" << std::endl;
        for (int i = 0; i < 5; ++i) outfile << r6.xval[i] << ", ";
        outfile << r6.xval[5] << std::endl;
        outfile << "Plane Equation" << std::endl;
        for (int i = 0; i < 3; ++i) outfile << plane[i] << ", ";
        outfile << plane[3] << std::endl;
        print_error(ctx, r6.xval, C, errorfile);                                                            // :464
        if (!in.config_out.empty() && ccal_write_config(config_path, in.config_out.c_str(), r6.xval) != CCAL_OK)
            throw Failure(CCAL_ERR_IO, "cannot write " + in.config_out);

        // ---- gradientDescentWithPlane (:466-524) --------------------------------------
        double x10[10];
        std::memcpy(x10, r6.xval, 6 * sizeof(double));
        for (int i = 0; i < 4; ++i) x10[6 + i] = plane[i];                                                  // :503-504
        ccal_gd_result r10;
        Check{ctx}(ccal_gd_minimize(ctx, CCAL_DOWNSAMPLED, 10, x10, &in.gd, &r10), "gd_minimize");
        outfile << "Gradient Descent on Plane..." << std::endl;
        outfile << "Iterations: " << r10.iterations << " Converged: " << (r10.converged ? "true" : "false")
                << " Final fval: " << r10.fval << std::endl;
        outfile << "Flange Transformation" << std::endl;
        for (int i = 0; i < 5; ++i) outfile << r10.xval[i] << ", ";
        outfile << r10.xval[5] << std::endl;
        outfile << "Plane Equation" << std::endl;
        for (int i = 0; i < 3; ++i) outfile << r10.xval[6 + i] << ", ";
        outfile << r10.xval[9] << std::endl;
        print_error(ctx, r10.xval, C, errorfile);                                                           // :523 (original plane)

        outfile << "------------------------------------------------------" << std::endl;                 // :579
        cleanup();
        return CCAL_OK;
    } catch (const std::exception& e) {
        if (comm && !exchanged && !ctxs.empty()) {
            // the peers are (or will be) waiting in the exchange: join it with the failure mark, so that
            // they return an error instead of waiting for a record that never comes
            Deadline d(comm_timeout_s, "the arg-min exchange (reporting a failure)");
            double c;
            int64_t i;
            (void)ccal_argmin_allgather(ctxs[0], comm, 1e9, CCAL_SHARD_FAILED, &c, &i, nullptr);
        }
        cleanup();
        if (errbuf && errbuf_len) {
            std::strncpy(errbuf, e.what(), errbuf_len - 1);
            errbuf[errbuf_len - 1] = 0;
        }
        // device trouble, bad arguments and capacity keep their status; everything that went wrong
        // while reading or writing files (the reference's bare `throw;`s) is an I/O error
        if (const Failure* f = dynamic_cast<const Failure*>(&e)) return f->status;
        return CCAL_ERR_IO;
    }
}

// reference TODO.md:5: the input config with the refined transformation written back
extern "C" int ccal_write_config(const char* config_in, const char* config_out, const double pose[6])
{
    if (!config_in || !config_out || !pose) return CCAL_ERR_ARG;
    try {
        std::string xml = slurp(config_in);
        const std::string open_tag = "<approximate_transformation>", close_tag = "</approximate_transformation>";
        const size_t b = xml.find(open_tag), e = xml.find(close_tag);
        if (b == std::string::npos || e == std::string::npos || e < b)
            throw std::runtime_error("config has no approximate_transformation element");
        // keep the element's indentation
        size_t line = xml.rfind('\n', b);
        std::string indent = line == std::string::npos ? "" : xml.substr(line + 1, b - line - 1);
        std::ostringstream body;
        body.precision(17);
        body << open_tag << "\n" << indent << "    <angle>rad</angle>\n" << indent << "    <metric>m</metric>\n"
             << indent << "    <value>";
        for (int i = 0; i < 6; ++i) body << pose[i] << (i < 5 ? "," : "");
        body << "</value>\n" << indent;
        xml.replace(b, e - b, body.str());
        std::ofstream f(config_out, std::ios::binary);
        if (!f) throw std::runtime_error("cannot write the output config");
        f << xml;
        return CCAL_OK;
    } catch (const std::exception&) {
        return CCAL_ERR_IO;
    }
}

// Loader entry points of the C ABI (thin wrappers; see include/ccal.h)
extern "C" int ccal_read_point_cloud(const char* filename, const char* metric, float** xyz_out, size_t* n_out)
{
    if (!filename || !xyz_out || !n_out) return CCAL_ERR_ARG;
    try {
        Cloud c = read_point_cloud(filename, metric ? metric : "m");
        float* p = (float*)std::malloc(std::max<size_t>(c.size(), 1) * sizeof(float));
        if (!p) return CCAL_ERR_IO;
        std::memcpy(p, c.data(), c.size() * sizeof(float));
        *xyz_out = p;
        *n_out = c.size() / 3;
        return CCAL_OK;
    } catch (const std::exception&) {
        *xyz_out = nullptr;
        *n_out = 0;
        return CCAL_ERR_IO;
    }
}

extern "C" int ccal_voxel_grid(const float* xyz, size_t n, float leaf, float* xyz_out, size_t* n_out)
{
    if ((!xyz && n) || !xyz_out || !n_out || !(leaf > 0)) return CCAL_ERR_ARG;
    Cloud in(xyz, xyz + 3 * n);
    Cloud out = voxel_grid(in, leaf);
    std::memcpy(xyz_out, out.data(), out.size() * sizeof(float));
    *n_out = out.size() / 3;
    return CCAL_OK;
}

extern "C" int ccal_filter_z(const float* xyz, size_t n, double z_max, float* xyz_out, size_t* n_out)
{
    if ((!xyz && n) || !xyz_out || !n_out) return CCAL_ERR_ARG;
    const Cloud kept = filter_z(Cloud(xyz, xyz + 3 * n), z_max);   // the routine the run itself uses
    std::memcpy(xyz_out, kept.data(), kept.size() * sizeof(float));
    *n_out = kept.size() / 3;
    return CCAL_OK;
}

extern "C" int ccal_parse_pose(const char* value_csv, const char* metric, const char* angle, double pose[6])
{
    if (!value_csv || !pose) return CCAL_ERR_ARG;
    for (int k = 0; k < 6; ++k) pose[k] = 0.0;
    const std::string a = angle ? angle : "radian";
    const int n = parse_pose_csv(value_csv, metric_scale(metric ? metric : "m"), a == "degree", pose);
    return n == 6 ? CCAL_OK : CCAL_ERR_IO;
}

extern "C" void ccal_free(void* p) { std::free(p); }
