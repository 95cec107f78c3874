// coati-alignpair: the `coati alignpair` verb (src/coati-alignpair.cc:30-50) for the
// marginal models, with the DP on an MI355X.
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <fcntl.h>
#include <iostream>
#include <spawn.h>
#include <stdexcept>
#include <string>
#include <sys/wait.h>
#include <thread>
#include <unistd.h>
#include <vector>

#include "cli.hpp"
#include "coati_hip.h"
#include "io.hpp"

extern char** environ;

namespace {
// --devices d0,d1,...: ONE PROCESS PER GPU.  This process only launches and waits -- it never touches
// HIP -- so every rank is a fresh process that initialises exactly one device (children are started
// with posix_spawn of this very binary plus the internal --dist-* flags; no exec from a process that has
// used the GPU).  Exit status: the first non-zero status of a rank, or 128 + signal.
int launch_ranks(const std::vector<int>& devices, int argc, char* argv[], const std::string& output) {
    const int world = static_cast<int>(devices.size());
    // the rendezvous id travels through a file in a directory only this user can enter (mkdtemp: mode 0700), so
    // that nobody else can plant the file -- or a symlink under the name rank 0 writes -- in /tmp
    char id_dir[] = "/tmp/coati-dist-XXXXXX";
    if(mkdtemp(id_dir) == nullptr) throw std::runtime_error("--devices: cannot create a rendezvous directory in /tmp");
    const std::string id_path_s = std::string(id_dir) + "/id";
    const char* const id_path = id_path_s.c_str();  // rank 0 creates it (by rename) once the id exists
    char self[4096];
    const ssize_t len = readlink("/proc/self/exe", self, sizeof self - 1);
    if(len <= 0) throw std::runtime_error("--devices: cannot resolve /proc/self/exe");
    self[len] = '\0';
    std::vector<pid_t> pids;
    for(int r = 0; r < world; ++r) {
        std::vector<std::string> av;
        av.emplace_back(self);
        for(int i = 1; i < argc; ++i) {
            const std::string a = argv[i];
            if(a == "--devices" || a == "--device") {  // replaced below
                ++i;
                continue;
            }
            av.push_back(a);
        }
        av.insert(av.end(), {"--device", std::to_string(devices[static_cast<std::size_t>(r)]), "--dist-rank", std::to_string(r),
                             "--dist-world", std::to_string(world), "--dist-id", id_path});
        std::vector<char*> cav;
        for(std::string& x : av) cav.push_back(x.data());
        cav.push_back(nullptr);
        pid_t pid = 0;
        posix_spawn_file_actions_t fa;
        posix_spawn_file_actions_init(&fa);
        if(r != 0) posix_spawn_file_actions_addopen(&fa, STDOUT_FILENO, "/dev/null", O_WRONLY, 0);  // only rank 0 writes the result
        const int rc = posix_spawn(&pid, self, &fa, nullptr, cav.data(), environ);
        posix_spawn_file_actions_destroy(&fa);
        if(rc != 0) {
            for(pid_t p : pids) kill(p, SIGTERM);
            throw std::runtime_error("--devices: posix_spawn failed");
        }
        pids.push_back(pid);
    }
    int status_all = 0;
    for(std::size_t left = pids.size(); left > 0; --left) {
        int st = 0;
        const pid_t done = waitpid(-1, &st, 0);
        if(done < 0) break;
        const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
        if(code != 0 && status_all == 0) {
            status_all = code;
            for(pid_t p : pids)  // a rank failed: the others would wait for it in a collective
                if(p != done) kill(p, SIGTERM);
        }
    }
    std::remove(id_path);
    std::remove((id_path_s + ".tmp").c_str());
    for(int r = 0; r < world; ++r) std::remove((id_path_s + ".part" + std::to_string(r)).c_str());  // (output on stdout: a failed run's slices)
    rmdir(id_dir);
    if(status_all != 0 && !(output.empty() || output == "-")) {
        // -o: the ranks assemble the result in <output>.tmp (renamed onto the output by rank 0 once every slice is in
        // place) from slices in <output>.partN; ranks that were stopped by the SIGTERM above could not tidy up after themselves
        const std::string final_path = coati_amd::extract_file_type(output).path;
        std::remove((final_path + ".tmp").c_str());
        for(int r = 0; r < world; ++r) std::remove((final_path + ".part" + std::to_string(r)).c_str());
    }
    return status_all;
}
}  // namespace

int main(int argc, char* argv[]) {
    using namespace coati_amd;
    // The HIP runtime takes 0.06-0.3 s to come up in a fresh process (it is most of a 10 000-pair run): for a batch run on
    // one device start it NOW, on a thread of its own, before the arguments are parsed and the model's matrix exponentials
    // are computed.  (Not in the --devices launcher, which must never touch HIP, see launch_ranks.)
    // (joined on every return path: a detached thread still inside the runtime's bring-up while main() returns -- a parse
    // error, --help -- would race the static destructors.  The batch path leaves through _Exit after the device has been used.)
    struct warm_up_t {
        std::thread t;
        ~warm_up_t() {
            if(t.joinable()) t.join();
        }
    } warm;
    {
        bool batch = false, launcher = false;
        for(int i = 1; i < argc; ++i) {
            const std::string a = argv[i];
            batch = batch || a == "--batch";
            launcher = launcher || a == "--devices";
        }
        if(batch && !launcher) warm.t = std::thread([] { (void)coati_hip_device_count(); });
    }
    args_t args;
    try {
        args = parse_arguments(verb_t::alignpair, argc, argv);
    } catch(const std::exception& e) {
        std::cerr << e.what() << "\nRun with --help for more information." << std::endl;
        return 106;  // CLI11's exit code for parse errors is non-zero; the exact value is not relied upon
    }
    if(args.help) {
        std::cout << usage(verb_t::alignpair);
        return EXIT_SUCCESS;
    }
    try {
        if(!args.aln.is_marginal()) {
            // the triplet/FST models (tri-mg, tri-ecm, dna) are a different algorithm and not served here
            throw std::invalid_argument("Mutation model unknown.");
        }
        if(args.dist_rank >= 0)  // a child of the --devices launcher below: rank dist_rank on --device
            return marg_alignment_batch_dist(args.aln, args.dist_rank, args.dist_world, args.dist_id) ? EXIT_SUCCESS : EXIT_FAILURE;
        if(args.devices.size() > 1) return launch_ranks(args.devices, argc, argv, args.aln.output);
        if(args.devices.size() == 1) args.aln.device = args.devices[0];
        if(args.batch) {
            // this process ends with the call: the GBs of HBM workspace the library cached go back with the process,
            // and so does the HIP runtime (tearing both down in order costs a 0.3 s run ~0.1 s)
            set_process_exits_after_call(true);
            const bool ok = marg_alignment_batch(args.aln);
            std::cout.flush();
            std::cerr.flush();
            std::_Exit(ok ? EXIT_SUCCESS : EXIT_FAILURE);
        }
        const bool ok = marg_alignment(args.aln);
        return ok ? EXIT_SUCCESS : EXIT_FAILURE;
    } catch(const std::exception& e) {
        std::cerr << "ERROR: " << e.what() << std::endl;
        return EXIT_FAILURE;
    }
}
