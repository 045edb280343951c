classdef xmArray < handle
% XMARRAY  Opaque handle to a single-precision tensor in MI355X device memory.
%
%   MATLAB's gpuArray is CUDA-only.  On an MI355X host the reference's
%       im = gpuArray(im) ;  ...  logits = gather(dag.vars(end).value) ;
%   (getBatchEmoVoxCeleb.m:197-205, fetch_emovoxceleb_imdb.m:129-131) become
%       im = xmArray(im) ;   ...  logits = gather(dag.vars(end).value) ;
%   and every vl_nn* gateway in mex/ accepts / returns xmArray objects (zero copy).  dagnn itself only moves
%   arrays around and calls vl_nn*; the places where it does arithmetic on gpuArrays directly are
%       dagnn.Sum.forward           (inputs{1} + inputs{2})      -> plus() below
%       cnn_train_dag/accumulateGradients                        -> xm_device('sgd' | 'average', ...)
%       ParameterServer push / sync                              -> xm_device('push' | 'sync_params')
%   dagnn.DagNN.move('gpu') calls gpuArray() on every parameter: put a one-line gpuArray.m shim
%   (function y = gpuArray(x), y = xmArray(x) ; end) in front of the toolbox on the path, or replace the call.
%
%   NOT EXERCISED IN THIS REPOSITORY (no MATLAB in the build image): documented binding, see INTEGRATION.md.
  properties (SetAccess = private)
    ptr = uint64(0)      % device address (xm_device_alloc)
    sz = [0 0 1 1]       % MATLAB size, padded to 4
  end
  methods
    function obj = xmArray(a, sz)
      if nargin == 2 && isa(a, 'uint64')      % adopt a buffer produced by a gateway
        obj.ptr = a ; obj.sz = sz ;
      elseif nargin == 1 && isa(a, 'xmArray')
        obj = a ;
      elseif nargin == 1
        obj = xm_device('upload', single(a)) ;
      end
    end
    function a = gather(obj), a = xm_device('download', obj) ; end
    function varargout = size(obj, d)
      s = obj.sz ; while numel(s) > 2 && s(end) == 1, s(end) = [] ; end
      if nargin > 1, s = obj.sz(min(d, 4)) ; if d > 4, s = 1 ; end, end
      if nargout <= 1, varargout{1} = s ; else
        for i = 1:nargout, if i <= numel(obj.sz), varargout{i} = obj.sz(i) ; else, varargout{i} = 1 ; end, end
      end
    end
    function n = numel(obj), n = prod(obj.sz) ; end
    function tf = isempty(obj), tf = prod(obj.sz) == 0 ; end
    function c = classUnderlying(~), c = 'single' ; end
    function y = plus(a, b), y = xm_device('sum', xmArray(a), xmArray(b)) ; end
    function delete(obj)
      if obj.ptr ~= 0, xm_device('free', obj.ptr) ; obj.ptr = uint64(0) ; end
    end
  end
end
