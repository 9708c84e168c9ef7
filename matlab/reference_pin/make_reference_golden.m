function make_reference_golden(refdir, outfile, repeats)
% MAKE_REFERENCE_GOLDEN  Pin the MI355X port to the reference itself.
%
%   make_reference_golden('/path/to/TFT_vs_Fund')            % writes reference_golden.mat next to this file
%   make_reference_golden(refdir, outfile, repeats)
%
% Runs the REFERENCE's own .m files (LauraFJulia/TFT_vs_Fund, taken from the checkout
% `refdir` -- nothing of it is shipped here) on the inputs of this repository's
% committed parity fixtures (reference_inputs.mat, written by
% tools/export_reference_inputs.py) and stores, per case and method,
%     R_t_2, R_t_3, Reconst, T, iter      the five outputs of  Method(Corresp,CalM)
%     seconds                             best tic/toc of `repeats` calls (default 3)
%     err                                 the error message if the call threw
% plus the MATLAB release, the machine and the thread count.  Copy the result to
% tests/golden/reference_golden.mat: tests/test_reference_golden.py then compares
% the numpy oracle (CPU suite) and the HIP kernels (pytest -m gpu) with it, and
% bench.py reports the timings as `cpu_baseline_reference`.
%
% It needs exactly what the reference needs (README.txt:33-36): base MATLAB for the
% eight pose methods (svd, pinv, null, rank); no toolbox.  Octave runs it as well.
% The wrappers of matlab/*.m (the MEX drop-ins) must NOT be on the path: the script
% removes this repository's matlab/ directory from it and checks where each method
% resolves.
if nargin < 1 || isempty(refdir), error('usage: make_reference_golden(refdir[, outfile[, repeats]])'); end
here = fileparts(mfilename('fullpath'));
if nargin < 2 || isempty(outfile), outfile = fullfile(here, 'reference_golden.mat'); end
if nargin < 3 || isempty(repeats), repeats = 3; end

dropins = fileparts(here);                                % <repo>/matlab holds same-named MEX wrappers
p = strsplit(path, pathsep);
for k = 1:numel(p)
    if strcmp(p{k}, dropins), rmpath(dropins); end
end
addpath(fullfile(refdir, 'TFT_methods'), fullfile(refdir, 'F_methods'), ...
        fullfile(refdir, 'auxiliar_functions'), fullfile(refdir, 'Optimization'), '-begin');

S = load(fullfile(here, 'reference_inputs.mat'));
cases = S.cases;
all_methods = {'LinearTFTPoseEstimation', 'ResslTFTPoseEstimation', 'NordbergTFTPoseEstimation', ...
               'FaugPapaTFTPoseEstimation', 'PiPoseEstimation', 'PiColPoseEstimation', ...
               'LinearFPoseEstimation', 'OptimFPoseEstimation'};
for m = 1:numel(all_methods)
    w = which(all_methods{m});
    if isempty(w) || isempty(strfind(w, refdir))
        error('%s resolves to "%s", not to the reference checkout %s', all_methods{m}, w, refdir);
    end
end

results = cell(numel(cases), 1);
for k = 1:numel(cases)
    c = cases(k);
    methods = c.methods;
    if ischar(methods), methods = {methods}; end
    r = struct('name', c.name, 'N', size(c.Corresp, 2));
    for m = 1:numel(methods)
        name = strtrim(methods{m});
        f = str2func(name);
        o = struct('R_t_2', [], 'R_t_3', [], 'Reconst', [], 'T', [], 'iter', NaN, 'seconds', NaN, 'err', '');
        best = Inf;
        try
            for rep = 1:repeats
                t0 = tic;
                [R_t_2, R_t_3, Reconst, T, iter] = f(c.Corresp, c.CalM);     % experiments.m:108
                best = min(best, toc(t0));
            end
            o.R_t_2 = R_t_2; o.R_t_3 = R_t_3; o.Reconst = Reconst; o.T = T; o.iter = double(iter); o.seconds = best;
        catch e
            o.err = e.message;
        end
        r.(name) = o;
        fprintf('%-40s %-28s N=%4d  %s\n', c.name, name, r.N, tern(isempty(o.err), sprintf('%.4f s, iter %g', o.seconds, o.iter), ['ERROR ' o.err]));
    end
    results{k} = r;
end

info = struct();
info.release = version();
info.is_octave = exist('OCTAVE_VERSION', 'builtin') ~= 0;
info.computer = computer();
info.date = datestr(now, 31);
try, info.threads = maxNumCompThreads(); catch, info.threads = NaN; end
try, info.cores = feature('numcores'); catch, info.cores = NaN; end
info.repeats = repeats;
info.refdir = refdir;
format_version = 1;
save(outfile, 'results', 'info', 'format_version', '-v7');
fprintf('wrote %s (%d cases).  Copy it to tests/golden/reference_golden.mat\n', outfile, numel(cases));
end

function s = tern(c, a, b)
if c, s = a; else, s = b; end
end
